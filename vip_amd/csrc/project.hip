// project.hip -- the two skinny GEMMs of `_project_subtract` (psfsub/pca_fullfr.py:1727-1731):
//     transformed   = V . M^T            (k x n)
//     reconstructed = transformed^T . V  (n x P)
//     residuals     = M - reconstructed
// restated through the Gram identity V = S^-1 E^T M  =>  reconstructed = E (E^T M):
//     rowspace_gemm :  T[k,P] = W[k,n] . M[n,P]        (W = E^T ; PCs = S^-1 T)
//     subtract_gemm :  R[n,P] = M[n,P] - C[n,k] . T[k,P]  (C = E ; optional recon output)
// Both stream M once from HBM (AI = k/2 flop/B -> HBM-bound) and keep the small operand in
// registers / L1.  MFMA: v_mfma_f32_32x32x2_f32 (exact f32 FMA chains), one wave per 128-pixel
// tile.  The streamed operand is loaded as float4 along the pixel axis: lane (j = lane&31,
// kh = lane>>5) loads M[f0+kh][px0 + 4j .. 4j+3]; component c feeds MFMA number c whose output
// column j is pixel px0 + 4j + c, so four 32x32 accumulators cover 128 contiguous pixels with
// fully coalesced 512-byte row segments and no LDS transpose.
#include <algorithm>
#include "common.h"

namespace vipmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same four floats at 4-byte alignment: ONE global_load / global_store_dwordx4 that the hardware splits where it crosses a
// line (unaligned access mode of the HSA target) -- rows of an odd-sized frame (P = 511 x 511: the reference's own convention) start
// at every alignment, and four scalar accesses per lane cost the projection kernels 1.3-1.5x (r06_kernel_stats_odd.csv)
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

template <bool VEC>
__device__ __forceinline__ f32x4 ldrow4(const float* __restrict__ base, int64_t row, int64_t nrows,
                                        int64_t P, int64_t px) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (row >= nrows || px >= P) return v;
  const float* p = base + row * P + px;
  if (VEC && px + 4 <= P) {
    v = *reinterpret_cast<const f32x4*>(p);
  } else if (px + 4 <= P) {
    v = *reinterpret_cast<const f32x4_u*>(p);
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (px + c < P) v[c] = p[c];
  }
  return v;
}

template <bool VEC>
__device__ __forceinline__ void strow4(float* __restrict__ base, int64_t row, int64_t nrows, int64_t P,
                                       int64_t px, f32x4 v) {
  if (row >= nrows || px >= P) return;
  float* p = base + row * P + px;
  if (VEC && px + 4 <= P) {
    *reinterpret_cast<f32x4*>(p) = v;
  } else if (px + 4 <= P) {
    *reinterpret_cast<f32x4_u*>(p) = v;
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (px + c < P) p[c] = v[c];
  }
}

// T[k,P] = Wt[n,kld]^T . M[n,P];  Wt row f holds W[0..k)[f] (kld >= 32*groups, zero padded).
// NG groups of 32 output rows per wave: with more than 32 components (C5: k = 50) one pass over M serves 64 of them
// (the streamed operand is the expensive one: 8.4 GB at C5; same accumulation order per output element as NG = 1).
template <bool VEC, int NG>
__global__ __launch_bounds__(256, 2) void rowspace_kernel(const float* __restrict__ Wt, int kld,
                                                       const float* __restrict__ M, int k, int n,
                                                       int64_t P, const float* __restrict__ rowscale,
                                                       float* __restrict__ T, int64_t sWt, int64_t sM, int64_t sT,
                                                       const int* __restrict__ frange = nullptr) {
  // frange (optional): [2 g], [2 g + 1] = the rows f of M outside which EVERY coefficient of output group g (32 rows of T) is zero,
  // as multiples of 8 -- the (I - C) A product of annular PCA, whose row j only has coefficients inside its library window: the
  // rows of M outside the union of the wave's groups are neither loaded nor multiplied (bit-identical: they would have added zeros)
  Wt += blockIdx.z * sWt;                    // blockIdx.z = problem of the batch (strides 0 for a single one)
  M += blockIdx.z * sM;
  T += blockIdx.z * sT;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  const int64_t px0 = tile * 128;
  if (px0 >= P) return;
  const int grp0 = blockIdx.y * NG;         // first group of 32 output rows
  const int jl = lane & 31, kh = lane >> 5;
  const int64_t px = px0 + 4 * jl;
  f32x16 acc[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][c][r] = 0.f;
  const float* wrow = Wt + grp0 * 32 + jl;  // A operand: W[grp*32 + i][f], i = lane&31
  constexpr int U = 4;
  int flo = 0, fhi = n;
  if (frange) {
    flo = n;
    fhi = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if ((grp0 + g) * 32 < k) {
        const int lo = frange[2 * (grp0 + g)], hi = frange[2 * (grp0 + g) + 1];
        flo = lo < flo ? lo : flo;
        fhi = hi > fhi ? hi : fhi;
      }
    fhi = fhi < n ? fhi : n;
  }
  for (int f0 = flo; f0 < fhi; f0 += 2 * U) {
    f32x4 b[U];
    float a[NG][U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = f0 + 2 * u + kh;
      b[u] = ldrow4<VEC>(M, f, n, P, px);
#pragma unroll
      for (int g = 0; g < NG; ++g) a[g][u] = (f < n && (grp0 + g) * 32 < kld) ? wrow[(int64_t)f * kld + 32 * g] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[g][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][u], b[u][c], acc[g][c], 0, 0, 0);
  }
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (grp0 + g) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (i < k) {
        const float s = rowscale ? rowscale[i] : 1.f;
        f32x4 v = {acc[g][0][r] * s, acc[g][1][r] * s, acc[g][2][r] * s, acc[g][3][r] * s};
        strow4<VEC>(T, i, k, P, px, v);
      }
    }
}

// The (I - C) A product of annular PCA for ALL segments in one launch (round 6): M holds the segment matrices side by side
// ([n][Ptot], every segment a whole number of 128-pixel tiles), tile -> segment through `tile_seg`, the n x n coefficient matrix of
// the tile's segment at Wt_all + seg * n * kld (transposed layout, as rowspace_kernel reads it), its library windows at
// frange_all + seg * 2 * groups.  The result is not written as a matrix: row i of a tile goes straight to frame i of the output cube
// THROUGH the pixel list (pix_out[p] = flat pixel index of column p, -1 = padding or a pixel a later segment owns) -- the scatter
// pass, the residual matrix and seven of eight launches disappear.  Four consecutive list entries are four consecutive pixels of an
// image row almost everywhere: one 16-byte store when the address allows, 8 + 8 or 4 + 8 + 4 otherwise, single pixels at the ends
// of a run.  Same accumulation order per element as rowspace_kernel<true, 2>: bit-identical residuals.
__global__ __launch_bounds__(256, 2) void rowspace_scatter_kernel(const float* __restrict__ Wt_all, int kld, const float* __restrict__ M,
                                                                int n, int64_t Ptot, const int32_t* __restrict__ tile_seg,
                                                                const int32_t* __restrict__ pix_out, const int* __restrict__ frange_all,
                                                                int64_t P, float* __restrict__ out,
                                                                const float* __restrict__ rho_all, const float* __restrict__ mu32) {
  // rho_all / mu32 (optional: a float64 cube whose per-pixel temporal mean is carried apart, pca_f64.hip): M is D = cube - 1 mu^T and
  // the residuals are (I - C) D + rho mu^T, rho[seg * n + i] the row sums of I - C: one multiply-add per element on the way out
  constexpr int NG = 2;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  const int64_t px0 = tile * 128;
  if (px0 >= Ptot) return;
  const int seg = __builtin_amdgcn_readfirstlane(tile_seg[tile]);
  if (seg < 0) return;                                  // (a tile of padding only)
  const int groups = (n + 31) / 32;
  const float* Wt = Wt_all + (int64_t)seg * n * kld;
  const int* frange = frange_all ? frange_all + (int64_t)seg * 2 * groups : nullptr;
  const int grp0 = blockIdx.y * NG;
  const int jl = lane & 31, kh = lane >> 5;
  const int64_t px = px0 + 4 * jl;
  f32x16 acc[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][c][r] = 0.f;
  const float* wrow = Wt + grp0 * 32 + jl;
  constexpr int U = 4;
  int flo = 0, fhi = n;
  if (frange) {
    flo = n;
    fhi = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if ((grp0 + g) * 32 < n) {
        const int lo = frange[2 * (grp0 + g)], hi = frange[2 * (grp0 + g) + 1];
        flo = lo < flo ? lo : flo;
        fhi = hi > fhi ? hi : fhi;
      }
    fhi = fhi < n ? fhi : n;
  }
  for (int f0 = flo; f0 < fhi; f0 += 2 * U) {
    f32x4 b[U];
    float a[NG][U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = f0 + 2 * u + kh;
      b[u] = ldrow4<true>(M, f, n, Ptot, px);
#pragma unroll
      for (int g = 0; g < NG; ++g) a[g][u] = (f < n && (grp0 + g) * 32 < kld) ? wrow[(int64_t)f * kld + 32 * g] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[g][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][u], b[u][c], acc[g][c], 0, 0, 0);
  }
  const int4 q = *reinterpret_cast<const int4*>(pix_out + px);
  const bool run4 = q.x >= 0 && q.y == q.x + 1 && q.z == q.x + 2 && q.w == q.x + 3;
  f32x4 m4 = {0.f, 0.f, 0.f, 0.f};
  if (rho_all) m4 = *reinterpret_cast<const f32x4*>(mu32 + px);
  const float* rho = rho_all ? rho_all + (int64_t)seg * n : nullptr;
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (grp0 + g) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (i < n) {
        float* row = out + (int64_t)i * P;
        float v0 = acc[g][0][r], v1 = acc[g][1][r], v2 = acc[g][2][r], v3 = acc[g][3][r];
        if (rho) {
          const float ri = rho[i];
          v0 = fmaf(ri, m4[0], v0);
          v1 = fmaf(ri, m4[1], v1);
          v2 = fmaf(ri, m4[2], v2);
          v3 = fmaf(ri, m4[3], v3);
        }
        if (run4) {
          float* d = row + q.x;
          const int al = (int)((reinterpret_cast<uintptr_t>(d) >> 2) & 3);
          if (al == 0) {
            *reinterpret_cast<f32x4*>(d) = f32x4{v0, v1, v2, v3};
          } else if (al == 2) {
            *reinterpret_cast<float2*>(d) = float2{v0, v1};
            *reinterpret_cast<float2*>(d + 2) = float2{v2, v3};
          } else {
            d[0] = v0;
            *reinterpret_cast<float2*>(d + 1) = float2{v1, v2};
            d[3] = v3;
          }
        } else {
          if (q.x >= 0) row[q.x] = v0;
          if (q.y >= 0) row[q.y] = v1;
          if (q.z >= 0) row[q.z] = v2;
          if (q.w >= 0) row[q.w] = v3;
        }
      }
    }
}

// A residual that cancels to EXACTLY 0 although its sample is not 0.  The reference's rotation masks by VALUE: pixels equal to
// `mask_val` are reset after the rotation (preproc/derotation.py:133-140,324-326), and pca(mask_center_px=...) runs it with
// mask_val = 0 (psfsub/pca_fullfr.py:412-415) -- in its float64 arithmetic only the masked disc is ever exactly 0.  float32
// residuals are quantised to the ulp of the SAMPLE (5e-4 on detector counts of 7e3, 1e-6 on the benchmark's cubes): one residual in
// ~1e5 .. 1e6 cancels to 0.0f, would be taken for a masked pixel and replace one of the n values under the median by 0 (found with
// the float64 golden g28: one pixel of the frame off by 1.3).  Such a residual becomes 1e-30 -- nothing to the shears, and no
// longer "equal to the mask value".  Masked samples (exactly 0, with every component 0 there) keep their exact 0; NaN stays NaN.
// OPT-IN per call (`guard`; context option `sub_guard`, default 1): median_sub's `cube - median` wants the exact zeros -- for an odd
// frame count np.median returns one of the samples, the reference's residual IS 0 there in float64 too, and its mask_val = 0 rotation
// resets those pixels (psfsub/medsub.py:279-285 with radius_int > 0): psfsub/medsub.py clears the option around its subtraction.
__device__ __forceinline__ f32x4 keep_nonzero(f32x4 res, f32x4 m, int guard) {
#pragma unroll
  for (int i = 0; i < 4; ++i) res[i] = (guard && res[i] == 0.f && m[i] != 0.f) ? 1e-30f : res[i];
  return res;
}

// R[n,P] = M - Ct[k,nld]^T . T[k,P];  Ct row c holds C[0..n)[c] (nld >= n rounded to 32).
template <bool VEC, bool RECON>
__global__ __launch_bounds__(256) void subtract_kernel(const float* __restrict__ M,
                                                       const float* __restrict__ Ct, int nld,
                                                       const float* __restrict__ T, int n, int k,
                                                       int64_t P, float* __restrict__ R,
                                                       float* __restrict__ recon, int64_t sM, int64_t sCt, int64_t sT, int guard) {
  M += blockIdx.y * sM;                      // blockIdx.y = problem of the batch (strides 0 for a single one)
  R += blockIdx.y * sM;
  Ct += blockIdx.y * sCt;
  T += blockIdx.y * sT;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  const int64_t px0 = tile * 128;
  if (px0 >= P) return;
  const int jl = lane & 31, kh = lane >> 5;
  const int64_t px = px0 + 4 * jl;
  constexpr int KS = 16;                       // component pairs held in registers per chunk
  // up to 32 components: the wave's tile of T (its 128 pixels, every component) stays in registers for all frame blocks --
  // it was re-read from L1 / L2 for every block of 32 frames (13 times at C2)
  const bool hoist = k <= 2 * KS;
  f32x4 bh[KS];
  if (hoist) {
#pragma unroll
    for (int s = 0; s < KS; ++s) bh[s] = ldrow4<VEC>(T, 2 * s + kh, k, P, px);
  }
  for (int fb = 0; fb < n; fb += 32) {
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    f32x4 m[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = fb + (r & 3) + 8 * (r >> 2) + 4 * kh;
      m[r] = ldrow4<VEC>(M, f, n, P, px);
    }
    for (int c0 = 0; c0 < k; c0 += 2 * KS) {
      f32x4 b[KS];
      float a[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int comp = c0 + 2 * s + kh;
        b[s] = hoist ? bh[s] : ldrow4<VEC>(T, comp, k, P, px);
        a[s] = (comp < k) ? Ct[(int64_t)comp * nld + fb + jl] : 0.f;
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (c0 + 2 * s < k) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s][c], acc[c], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = fb + (r & 3) + 8 * (r >> 2) + 4 * kh;
      f32x4 rec = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
      f32x4 res = keep_nonzero(m[r] - rec, m[r], guard);
      strow4<VEC>(R, f, n, P, px, res);
      if (RECON) strow4<VEC>(recon, f, n, P, px, rec);
    }
  }
}

// The default subtraction (up to 128 components).  The FOUR waves of a workgroup share one 128-pixel tile of T, staged once in
// LDS ([component][128 pixels], <= 64 KB), and split the blocks of 32 frames between them.  subtract_kernel above keeps the
// tile in registers (up to 32 components) or re-reads it through the L2 per frame block (more: 13 GB at C5 beside the 16.8 GB
// of M and R), and with both operand sets needs 426 registers -- ONE wave per SIMD, whose load / MFMA / store phases nothing
// overlaps.  This one takes 228: two waves per SIMD.  C2 (k = 20) 213 -> 177 us, C5 (k = 50) 6.1 -> 4.6 ms; same accumulation
// order per element, bit-identical.
template <bool VEC, bool RECON>
__global__ __launch_bounds__(256, 2) void subtract_lds_kernel(const float* __restrict__ M, const float* __restrict__ Ct, int nld,
                                                           const float* __restrict__ T, int n, int k, int64_t P,
                                                           float* __restrict__ R, float* __restrict__ recon,
                                                           int64_t sM, int64_t sCt, int64_t sT, int guard) {
  extern __shared__ __attribute__((aligned(16))) float tsm[];          // [kp][128], kp = k rounded up to even
  M += blockIdx.y * sM;                      // blockIdx.y = problem of the batch (strides 0 for a single one)
  R += blockIdx.y * sM;
  Ct += blockIdx.y * sCt;
  T += blockIdx.y * sT;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t px0 = (int64_t)blockIdx.x * 128;
  const int jl = lane & 31, kh = lane >> 5;
  const int64_t px = px0 + 4 * jl;
  const int kp = (k + 1) & ~1;
  for (int e = threadIdx.x; e < kp * 32; e += 256) {
    const int comp = e >> 5, q = e & 31;
    const f32x4 v = ldrow4<VEC>(T, comp, k, P, px0 + 4 * q);
    *reinterpret_cast<f32x4*>(tsm + comp * 128 + 4 * q) = v;
  }
  __syncthreads();
  for (int fb = 32 * wave; fb < n; fb += 128) {
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    f32x4 m[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = fb + (r & 3) + 8 * (r >> 2) + 4 * kh;
      m[r] = ldrow4<VEC>(M, f, n, P, px);
    }
    constexpr int KS = 8;
    for (int c0 = 0; c0 < kp; c0 += 2 * KS) {
      f32x4 b[KS];
      float a[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int comp = c0 + 2 * s + kh;
        b[s] = comp < kp ? *reinterpret_cast<const f32x4*>(tsm + comp * 128 + 4 * jl) : f32x4{0.f, 0.f, 0.f, 0.f};
        a[s] = (comp < k) ? Ct[(int64_t)comp * nld + fb + jl] : 0.f;
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (c0 + 2 * s < kp) {
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s][c], acc[c], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = fb + (r & 3) + 8 * (r >> 2) + 4 * kh;
      f32x4 rec = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
      f32x4 res = keep_nonzero(m[r] - rec, m[r], guard);
      strow4<VEC>(R, f, n, P, px, res);
      if (RECON) strow4<VEC>(recon, f, n, P, px, rec);
    }
  }
}

// dst[cols, ldd] = src[rows, cols]^T, zero padded to ldd (blockIdx.y = matrix of a contiguous batch)
__global__ void transpose_pad_kernel(const float* __restrict__ src, int rows, int cols,
                                     float* __restrict__ dst, int ldd) {
  src += (int64_t)blockIdx.y * rows * cols;
  dst += (int64_t)blockIdx.y * cols * ldd;
  int64_t total = (int64_t)cols * ldd;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(e / ldd), r = (int)(e % ldd);
    dst[e] = (r < rows) ? src[(int64_t)r * cols + c] : 0.f;
  }
}

// dst[rows, ldd] = src[rows, cols], zero padded to ldd (blockIdx.y = matrix of a contiguous batch)
__global__ void pad_cols_kernel(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst, int ldd) {
  src += (int64_t)blockIdx.y * rows * cols;
  dst += (int64_t)blockIdx.y * rows * ldd;
  const int64_t total = (int64_t)rows * ldd;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / ldd), c = (int)(e % ldd);
    dst[e] = (c < cols) ? src[(int64_t)r * cols + c] : 0.f;
  }
}

}  // namespace

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// internal: operands already in the layouts the kernels want
int rowspace_gemm_t(vipmi_ctx* ctx, const float* Wt, int kld, const float* M, int64_t k, int64_t n,
                    int64_t P, const float* rowscale, float* T, const int* frange) {
  const bool vec = (P % 4 == 0) && aligned16(M) && aligned16(T);
  const int groups = (int)cdiv(k, 32), ng = groups >= 2 ? 2 : 1;     // two groups of 32 components per pass over M from 33 on
  dim3 grid((unsigned)cdiv(cdiv(P, 128), 4), (unsigned)cdiv(groups, ng)), block(256);
#define LAUNCH(V, G)                                                                                          \
  hipLaunchKernelGGL((rowspace_kernel<V, G>), grid, block, 0, ctx->stream, Wt, kld, M, (int)k, (int)n, P, rowscale, T, \
                     (int64_t)0, (int64_t)0, (int64_t)0, frange)
  if (vec) {
    if (ng == 2) LAUNCH(true, 2); else LAUNCH(true, 1);
  } else {
    if (ng == 2) LAUNCH(false, 2); else LAUNCH(false, 1);
  }
#undef LAUNCH
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

int subtract_gemm_t(vipmi_ctx* ctx, const float* M, const float* Ct, int nld, const float* T, int64_t n,
                    int64_t k, int64_t P, float* R, float* recon) {
  const bool vec = (P % 4 == 0) && aligned16(M) && aligned16(T) && aligned16(R) &&
                   (!recon || aligned16(recon));
  dim3 grid((unsigned)cdiv(cdiv(P, 128), 4)), block(256);
  const int guard = ctx->opt("sub_guard", 1) != 0;           // (see keep_nonzero)
  if (k <= 128 && ctx->opt("subtract_lds", 1) != 0) {      // (see subtract_lds_kernel)
    const size_t lds = (size_t)((k + 1) & ~(int64_t)1) * 128 * sizeof(float);
    dim3 g1((unsigned)cdiv(P, 128));
#define LAUNCHL(V, RC)                                                                                              \
  do {                                                                                                               \
    auto kern = subtract_lds_kernel<V, RC>;                                                                          \
    VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds)); \
    hipLaunchKernelGGL(kern, g1, block, lds, ctx->stream, M, Ct, nld, T, (int)n, (int)k, P, R, recon, (int64_t)0, (int64_t)0, (int64_t)0, guard); \
  } while (0)
    if (vec) {
      if (recon) LAUNCHL(true, true); else LAUNCHL(true, false);
    } else {
      if (recon) LAUNCHL(false, true); else LAUNCHL(false, false);
    }
#undef LAUNCHL
    VIPMI_CHECK_HIP(hipGetLastError());
    return VIPMI_OK;
  }
#define LAUNCH(V, RC)                                                                             \
  hipLaunchKernelGGL((subtract_kernel<V, RC>), grid, block, 0, ctx->stream, M, Ct, nld, T, (int)n, \
                     (int)k, P, R, recon, (int64_t)0, (int64_t)0, (int64_t)0, guard)
  if (vec) {
    if (recon) LAUNCH(true, true); else LAUNCH(true, false);
  } else {
    if (recon) LAUNCH(false, true); else LAUNCH(false, false);
  }
#undef LAUNCH
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

int rowspace_scatter_f32(vipmi_ctx* ctx, const float* Wt_all, int kld, const float* M, int64_t n, int64_t Ptot,
                         const int32_t* tile_seg, const int32_t* pix_out, const int* frange_all, int64_t P, float* out,
                         const float* rho_all, const float* mu32) {
  VIPMI_REQUIRE((rho_all == nullptr) == (mu32 == nullptr) && (!mu32 || aligned16(mu32)), "rowspace_scatter: rho / mu go together");
  VIPMI_REQUIRE(Wt_all && M && tile_seg && pix_out && out, "rowspace_scatter: null pointer");
  VIPMI_REQUIRE(n > 0 && Ptot > 0 && Ptot % 128 == 0 && aligned16(M) && aligned16(pix_out), "rowspace_scatter: bad sizes / alignment");
  StageScope sc(ctx, "project");
  const int groups = (int)cdiv(n, 32);
  dim3 grid((unsigned)cdiv(Ptot / 128, 4), (unsigned)cdiv(groups, 2));
  hipLaunchKernelGGL(rowspace_scatter_kernel, grid, dim3(256), 0, ctx->stream, Wt_all, kld, M, (int)n, Ptot, tile_seg, pix_out, frange_all, P,
                     out, rho_all, mu32);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

int rowspace_gemm_f32(vipmi_ctx* ctx, const float* W, const float* M, int64_t k, int64_t n, int64_t P,
                      const float* rowscale, float* B, const int* frange) {
  VIPMI_REQUIRE(W && M && B, "rowspace_gemm: null pointer");
  VIPMI_REQUIRE(k > 0 && n > 0 && P > 0, "rowspace_gemm: bad sizes");
  StageScope sc(ctx, "project");
  const int kld = (int)cdiv(k, 32) * 32;
  float* Wt = nullptr;
  VIPMI_TRY(ws(ctx, "proj_wt", (size_t)n * kld, &Wt));
  hipLaunchKernelGGL(transpose_pad_kernel, dim3(64), dim3(256), 0, ctx->stream, W, (int)k, (int)n, Wt, kld);
  VIPMI_CHECK_HIP(hipGetLastError());
  return rowspace_gemm_t(ctx, Wt, kld, M, k, n, P, rowscale, B, frange);
}

int subtract_gemm_f32(vipmi_ctx* ctx, const float* M, const float* C, const float* B, int64_t n,
                      int64_t k, int64_t P, float* R, float* recon) {
  VIPMI_REQUIRE(M && C && B && R, "subtract_gemm: null pointer");
  VIPMI_REQUIRE(k > 0 && n > 0 && P > 0, "subtract_gemm: bad sizes");
  StageScope sc(ctx, "project");
  const int nld = (int)cdiv(n, 32) * 32;
  float* Ct = nullptr;
  VIPMI_TRY(ws(ctx, "proj_ct", (size_t)k * nld, &Ct));
  hipLaunchKernelGGL(transpose_pad_kernel, dim3(64), dim3(256), 0, ctx->stream, C, (int)n, (int)k, Ct, nld);
  VIPMI_CHECK_HIP(hipGetLastError());
  return subtract_gemm_t(ctx, M, Ct, nld, B, n, k, P, R, recon);
}

// R[b] = M[b] - E[b]^T (E[b] M[b]) for a contiguous batch: M, R [nb][n][P], E [nb][k][n] (rows = the leading
// eigenvectors of M[b] M[b]^T, already zeroed where rejected).  Two launches per chunk of problems instead of four
// per problem: the spectral PCAs of ADI+mSDI (one per multispectral frame, pca_fullfr.py:1482-1520) and the channels
// of a 4-D cube (pca_fullfr.py:544-658) are launch-bound otherwise.
int project_batched_f32(vipmi_ctx* ctx, const float* M, const float* E, int64_t nb, int64_t n, int64_t k, int64_t P,
                        float* R) {
  VIPMI_REQUIRE(M && E && R, "project_batched: null pointer");
  VIPMI_REQUIRE(nb > 0 && k > 0 && n > 0 && P > 0 && k <= n, "project_batched: bad sizes");
  StageScope sc(ctx, "project");
  const int kld = (int)cdiv(k, 32) * 32, nld = (int)cdiv(n, 32) * 32;
  // chunk of problems per launch: T of at most ~256 MB, grid dimensions within 65535
  int64_t chunk = std::max<int64_t>(1, (int64_t)(256u << 20) / (int64_t)(k * P * sizeof(float)));
  chunk = std::min<int64_t>(std::min<int64_t>(chunk, nb), 65535);
  float *Wt = nullptr, *Ct = nullptr, *T = nullptr;
  VIPMI_TRY(ws(ctx, "projb_wt", (size_t)chunk * n * kld, &Wt));
  VIPMI_TRY(ws(ctx, "projb_ct", (size_t)chunk * k * nld, &Ct));
  VIPMI_TRY(ws(ctx, "projb_t", (size_t)chunk * k * P, &T));
  const bool vec = (P % 4 == 0) && aligned16(M) && aligned16(R) && aligned16(T);
  const bool use_lds = k <= 128 && ctx->opt("subtract_lds", 1) != 0;
  const int guard = ctx->opt("sub_guard", 1) != 0;
  const size_t lds_t = (size_t)((k + 1) & ~(int64_t)1) * 128 * sizeof(float);
  if (use_lds) {
    VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(subtract_lds_kernel<true, false>), (int)lds_t));
    VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(subtract_lds_kernel<false, false>), (int)lds_t));
  }
  for (int64_t b0 = 0; b0 < nb; b0 += chunk) {
    const int64_t cb = std::min(chunk, nb - b0);
    const float* Mb = M + b0 * n * P;
    const float* Eb = E + b0 * k * n;
    float* Rb = R + b0 * n * P;
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(8, (unsigned)cb), dim3(256), 0, ctx->stream, Eb, (int)k, (int)n, Wt, kld);
    hipLaunchKernelGGL(pad_cols_kernel, dim3(8, (unsigned)cb), dim3(256), 0, ctx->stream, Eb, (int)k, (int)n, Ct, nld);
    dim3 g1((unsigned)cdiv(cdiv(P, 128), 4), (unsigned)cdiv(k, 32), (unsigned)cb), g2((unsigned)cdiv(cdiv(P, 128), 4), (unsigned)cb), g3((unsigned)cdiv(P, 128), (unsigned)cb);
    if (vec) {
      hipLaunchKernelGGL((rowspace_kernel<true, 1>), g1, dim3(256), 0, ctx->stream, Wt, kld, Mb, (int)k, (int)n, P,
                         (const float*)nullptr, T, (int64_t)n * kld, (int64_t)n * P, (int64_t)k * P);
      if (use_lds)
        hipLaunchKernelGGL((subtract_lds_kernel<true, false>), g3, dim3(256), lds_t, ctx->stream, Mb, Ct, nld, T, (int)n, (int)k, P, Rb,
                           (float*)nullptr, (int64_t)n * P, (int64_t)k * nld, (int64_t)k * P, guard);
      else
        hipLaunchKernelGGL((subtract_kernel<true, false>), g2, dim3(256), 0, ctx->stream, Mb, Ct, nld, T, (int)n, (int)k, P, Rb,
                           (float*)nullptr, (int64_t)n * P, (int64_t)k * nld, (int64_t)k * P, guard);
    } else {
      hipLaunchKernelGGL((rowspace_kernel<false, 1>), g1, dim3(256), 0, ctx->stream, Wt, kld, Mb, (int)k, (int)n, P,
                         (const float*)nullptr, T, (int64_t)n * kld, (int64_t)n * P, (int64_t)k * P);
      if (use_lds)
        hipLaunchKernelGGL((subtract_lds_kernel<false, false>), g3, dim3(256), lds_t, ctx->stream, Mb, Ct, nld, T, (int)n, (int)k, P, Rb,
                           (float*)nullptr, (int64_t)n * P, (int64_t)k * nld, (int64_t)k * P, guard);
      else
        hipLaunchKernelGGL((subtract_kernel<false, false>), g2, dim3(256), 0, ctx->stream, Mb, Ct, nld, T, (int)n, (int)k, P, Rb,
                           (float*)nullptr, (int64_t)n * P, (int64_t)k * nld, (int64_t)k * P, guard);
    }
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  return VIPMI_OK;
}

}  // namespace vipmi
