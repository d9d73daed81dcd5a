// annular.hip -- per-frame library PCA of one annulus segment: the reference's
// `do_pca_patch` (psfsub/pca_local.py:830-909) for all n frames of a segment matrix at once.
//
// Reference: for every frame j, library rows lib_j = `_find_indices_adi(...)` of the segment matrix
// A (n x npx); V = top-k right singular vectors of A[lib_j]; residual_j = A[j] - (A[j] V^T) V.
// That is n thin SVDs per segment.  Sub-Gram identity: with G = A A^T,
//   H_j = G[lib_j, lib_j] = E L E^T      (m x m, m <= max_frames_lib)
//   c_j = E_k L_k^-1 E_k^T G[lib_j, j]   (coefficients of the library frames)
//   residual_j = A[j] - sum_i c_j[i] A[lib_j[i]]
// so ONE Gram per segment (matrix cores), n small eigenproblems solved by the batched block-Jacobi
// kernel, and one dense n x n x npx projection GEMM (matrix cores) reproduce the n SVDs exactly.
#include "common.h"

namespace vipmi {

namespace {

// H[j][a][b] = G[idx[j][a]][idx[j][b]] for a,b < len[j], 0 elsewhere (zero padding only adds null
// eigenvalues, sorted last)
__global__ __launch_bounds__(256) void subgram_kernel(const double* __restrict__ G, int n, const int32_t* __restrict__ idx,
                                                      const int32_t* __restrict__ len, int max_lib, int m, double* __restrict__ H) {
  // one workgroup per library: the index list once into LDS, then row a by wave (a = wave, wave + 4, ...), columns by lane
  // -- no integer division per element, 512-byte row segments of H per store instruction
  extern __shared__ int32_t sidx[];
  const int j = blockIdx.x;
  const int lj = len[j];
  const int32_t* ij = idx + (size_t)j * max_lib;
  double* Hj = H + (size_t)j * m * m;
  for (int a = threadIdx.x; a < lj; a += blockDim.x) sidx[a] = ij[a];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  for (int a = wave; a < m; a += nw) {
    const bool rowlive = a < lj;
    const double* Ga = G + (size_t)(rowlive ? sidx[a] : 0) * n;
    for (int b = lane; b < m; b += 64) Hj[(size_t)a * m + b] = (rowlive && b < lj) ? Ga[sidx[b]] : 0.0;
  }
}

// one workgroup per frame j: C[j][lib_j[a]] = sum_{c<k'} E[c][a] * (E[c] . g_j) / lambda_c, stored as D = I - C (the matrix is
// zeroed before the launch), so that the residuals are ONE product  R = D A  on the matrix cores
__global__ __launch_bounds__(256) void coeff_kernel(const double* __restrict__ G, int n,
                                                    const int32_t* __restrict__ idx,
                                                    const int32_t* __restrict__ len, int max_lib, int m,
                                                    const double* __restrict__ evals,
                                                    const double* __restrict__ evecs, int k, int npx,
                                                    float* __restrict__ C, float* __restrict__ rho,
                                                    const int32_t* __restrict__ kseg = nullptr, int ldt = 0,
                                                    const double* __restrict__ u = nullptr) {
  // blockIdx.y = segment of a batch (round 6: all annuli in one launch): G[seg][n][n], C[seg][n][ldt or n]; the per-library arrays
  // (idx, len, evals, evecs, rho) are indexed by seg * n + j; kseg (optional): the segment's number of components.
  // ldt > 0: the matrix is written TRANSPOSED with row length ldt -- element (row j, frame f) at C[f * ldt + j], the layout the
  // row-space kernels read (no transposition pass afterwards).
  // rho (optional): rho[j] = 1 - sum_a c_j[a], the row sum of I - C in float64: with A = D + 1 mu^T (a float64 cube whose per-pixel
  // temporal mean is carried apart, pca_f64.hip) the residuals are (I - C) D + rho mu^T.  u (optional, [seg][n]): the offset is
  // u mu^T (the spatial scalings of a float64 cube) and rho = (I - C) u.
  extern __shared__ double sh[];      // g[m] | proj[k]
  double* g = sh;
  double* proj = sh + m;
  const int j = blockIdx.x;
  const size_t jj = (size_t)blockIdx.y * n + j;
  G += (size_t)blockIdx.y * n * n;
  C += (size_t)blockIdx.y * n * (ldt > 0 ? ldt : n);
  if (u) u += (size_t)blockIdx.y * n;
  if (kseg) k = kseg[blockIdx.y];
  const int lj = len[jj];
  const int32_t* ij = idx + jj * max_lib;
  const double* ev = evals + jj * m;
  const double* E = evecs + jj * m * m;
  int kk = k < lj ? k : lj;                    // get_eigenvectors: min(ncomp, min(shape)), svd.py:696
  if (kk > npx) kk = npx;
  for (int a = threadIdx.x; a < m; a += blockDim.x) g[a] = (a < lj) ? G[(size_t)ij[a] * n + j] : 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const double thr = ev[0] * 1e-12;
  for (int c = wave; c < kk; c += nw) {
    double s = 0;
    for (int a = lane; a < lj; a += 64) s += E[(size_t)c * m + a] * g[a];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) proj[c] = (ev[c] > thr && ev[c] > 0) ? s / ev[c] : 0.0;
  }
  __syncthreads();
  double csum = 0.0;
  for (int a = threadIdx.x; a < lj; a += blockDim.x) {
    double s = 0;
    for (int c = 0; c < kk; ++c) s += E[(size_t)c * m + a] * proj[c];
    if (ldt > 0) C[(size_t)ij[a] * ldt + j] = (float)(-s);
    else C[(size_t)j * n + ij[a]] = (float)(-s);
    csum += u ? s * u[ij[a]] : s;
  }
  __syncthreads();                       // (the frame may belong to its own library: add the identity afterwards)
  if (threadIdx.x == 0) C[ldt > 0 ? (size_t)j * ldt + j : (size_t)j * n + j] += 1.0f;
  if (rho) {                             // (uniform) fixed-order sum of the 256 partial sums
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) csum += __shfl_xor(csum, off, 64);
    __syncthreads();
    if (lane == 0) sh[wave] = csum;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int w = 0; w < nw; ++w) t += sh[w];
      rho[jj] = (float)((u ? u[j] : 1.0) - t);
    }
  }
}

// R[j, p] += rho[j] mu[p]
__global__ void rank1_add_kernel(float* __restrict__ R, const float* __restrict__ rho, const float* __restrict__ mu, int64_t n, int64_t P) {
  const int64_t total = n * P;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    R[e] += rho[e / P] * mu[e % P];
}

// frange[2 g], [2 g + 1]: the frames outside which row group g (32 rows) of D = I - C has no coefficient, as multiples of 8 -- the
// union over its rows j of [min(lib_j), max(lib_j)] and j itself.  One workgroup of 16 waves per group, two rows per wave, the lanes
// stride over a row's library (coalesced; one thread per row walking its 200 indices took 32 us per segment, one wave walking the 32
// rows one after the other 50: every row starts with a dependent load of its length).
__global__ __launch_bounds__(1024) void coeff_range_kernel(int n, const int32_t* __restrict__ idx, const int32_t* __restrict__ len,
                                                           int max_lib, int* __restrict__ frange) {
  __shared__ int slo[16], shi[16];
  const int g = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  idx += (size_t)blockIdx.y * n * max_lib;                 // blockIdx.y = segment of a batch
  len += (size_t)blockIdx.y * n;
  frange += (size_t)blockIdx.y * 2 * gridDim.x;
  int lo = n, hi = 0;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int j = g * 32 + 2 * wave + r;
    if (j < n) {
      if (lane == 0) {
        lo = j < lo ? j : lo;
        hi = j + 1 > hi ? j + 1 : hi;
      }
      const int32_t* ij = idx + (size_t)j * max_lib;
      const int lj = len[j];
      for (int a = lane; a < lj; a += 64) {
        const int v = ij[a];
        lo = v < lo ? v : lo;
        hi = v + 1 > hi ? v + 1 : hi;
      }
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int ol = __shfl_xor(lo, m, 64), oh = __shfl_xor(hi, m, 64);
    lo = ol < lo ? ol : lo;
    hi = oh > hi ? oh : hi;
  }
  if (lane == 0) {
    slo[wave] = lo;
    shi[wave] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) {
      lo = slo[w] < lo ? slo[w] : lo;
      hi = shi[w] > hi ? shi[w] : hi;
    }
    frange[2 * g] = lo & ~7;
    frange[2 * g + 1] = (hi + 7) & ~7;
  }
}

}  // namespace

// The three stages of a segment.  annular_residuals_multi_f32 below runs them back to back; the Python front runs stage 1
// for ALL segments, then ONE batched eigensolve over the libraries of all segments (3200 problems at C3: enough to run
// the tridiagonalisation and the rest of the solve as two launches with four problems per CU in the second), then
// stage 3 per segment.
// stage 1: G = A A^T (n x n) and the zero-padded sub-Gram matrices H[j] (m x m, m >= max_lib) of the n libraries
int annular_subgrams_f64(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* lib_idx,
                         const int32_t* lib_len, int64_t max_lib, int64_t m, double* G, double* H) {
  VIPMI_REQUIRE(A && lib_idx && lib_len && G && H, "annular_subgrams: null pointer");
  VIPMI_REQUIRE(n > 0 && npx > 0 && max_lib > 0 && max_lib <= n && m >= max_lib, "annular_subgrams: bad sizes");
  VIPMI_TRY(gram_f32(ctx, A, n, A, n, npx, npx, G));
  hipLaunchKernelGGL(subgram_kernel, dim3((unsigned)n), dim3(256), sizeof(int32_t) * (size_t)max_lib, ctx->stream, G, (int)n,
                     lib_idx, lib_len, (int)max_lib, (int)m, H);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

// stage 2 for ALL segments at once (round 5): the leading k eigenpairs of the nseg * n libraries, the solver gathering every
// sub-Gram matrix itself from its segment's Gram matrix G[seg] (n x n, L2-resident) -- H is never written out and read back
// (1 GB each way at C3, and the 1 ms of subgram_kernel launches).  lib_idx: [nseg * n][m] (rows padded to m), lib_len: [nseg * n];
// work[nseg * n][m][m]: workspace (the reflectors).  Sizes the register-resident solver does not serve (m > 200 or k > 32) take the
// old route: the matrices materialised into `work`, eigh_leading on them.
int annular_eigh_f64(vipmi_ctx* ctx, const double* G, int64_t nseg, int64_t n, const int32_t* lib_idx, const int32_t* lib_len,
                     int64_t m, int64_t k, double* work, double* evals, double* evecs) {
  VIPMI_REQUIRE(G && lib_idx && lib_len && work && evals && evecs, "annular_eigh: null pointer");
  VIPMI_REQUIRE(nseg > 0 && n > 0 && m > 0 && k > 0 && k <= m, "annular_eigh: bad sizes");
  if (eigh_gather_supported(m, k) && ctx->opt("eigh_reg", 1) != 0 && ctx->opt("ann_gather", 1) != 0)
    return eigh_topk_gather_f64(ctx, G, nseg, n, n, lib_idx, lib_len, m, k, work, evals, evecs);
  for (int64_t sg = 0; sg < nseg; ++sg) {
    hipLaunchKernelGGL(subgram_kernel, dim3((unsigned)n), dim3(256), sizeof(int32_t) * (size_t)m, ctx->stream, G + (size_t)sg * n * n, (int)n,
                       lib_idx + (size_t)sg * n * m, lib_len + (size_t)sg * n, (int)m, (int)m, work + (size_t)sg * n * m * m);
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  return eigh_leading(ctx, work, nseg * n, m, k, lib_len, evals, evecs);
}

// stage 3: residuals[i] = (I - C_i) A for every truncation rank ncomps[i] (HOST array), from the leading eigenpairs of the
// libraries (evals[j][m], evecs[j][m][m]: rows = vectors, as the top-k eigensolver returns them)
int annular_apply_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* lib_idx,
                      const int32_t* lib_len, int64_t max_lib, int64_t m, const double* G, const double* evals,
                      const double* evecs, const int32_t* ncomps, int64_t nk, float* residuals, const float* mu32, const double* u) {
  VIPMI_REQUIRE(A && lib_idx && lib_len && G && evals && evecs && ncomps && residuals, "annular_apply: null pointer");
  VIPMI_REQUIRE(!u || mu32, "annular_apply: offset weights without an offset");
  VIPMI_REQUIRE(n > 0 && npx > 0 && max_lib > 0 && m >= max_lib && nk > 0, "annular_apply: bad sizes");
  int64_t kmax = 0;
  for (int64_t i = 0; i < nk; ++i) {
    VIPMI_REQUIRE(ncomps[i] > 0, "annular_residuals: ncomp must be positive");
    if (ncomps[i] > kmax) kmax = ncomps[i];
  }
  float* C = nullptr;
  VIPMI_TRY(ws(ctx, "ann_C", (size_t)n * n, &C));
  // the frames each group of 32 rows of I - C reaches (its rows' library windows): the product skips the rest
  int* frange = nullptr;
  const int groups = (int)cdiv(n, 32);
  if (ctx->opt("ann_range", 1) != 0) {
    VIPMI_TRY(ws(ctx, "ann_frange", (size_t)2 * groups, &frange));
    hipLaunchKernelGGL(coeff_range_kernel, dim3(groups), dim3(1024), 0, ctx->stream, (int)n, lib_idx, lib_len, (int)max_lib, frange);
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  const size_t shm = (size_t)(m + kmax + 8) * sizeof(double);
  float* rho = nullptr;                  // mu32: A is D = M - 1 mu^T of a float64 cube; residuals = (I - C) D + rho mu^T
  if (mu32) VIPMI_TRY(ws(ctx, "ann_rho", (size_t)n, &rho));
  for (int64_t i = 0; i < nk; ++i) {
    VIPMI_CHECK_HIP(hipMemsetAsync(C, 0, sizeof(float) * n * n, ctx->stream));
    hipLaunchKernelGGL(coeff_kernel, dim3((unsigned)n), dim3(256), shm, ctx->stream, G, (int)n, lib_idx, lib_len,
                       (int)max_lib, (int)m, evals, evecs, (int)ncomps[i], (int)npx, C, rho, (const int32_t*)nullptr, 0, u);
    VIPMI_CHECK_HIP(hipGetLastError());
    // residuals = A - C A = (I - C) A: one (n x n) x (n x npx) product on the matrix cores.  (Round 1 ran it through the
    // skinny-k subtract kernel, k = n components: 14 TF/s, 4.7 of C3's 35 ms; the row-space kernel does it at ~60 TF/s.)
    VIPMI_TRY(rowspace_gemm_f32(ctx, C, A, n, n, npx, nullptr, residuals + (size_t)i * n * npx, frange));
    if (mu32) {
      const int64_t blocks = cdiv(n * npx, 2048);
      hipLaunchKernelGGL(rank1_add_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, ctx->stream,
                         residuals + (size_t)i * n * npx, rho, mu32, n, npx);
      VIPMI_CHECK_HIP(hipGetLastError());
    }
  }
  return VIPMI_OK;
}

// ---- round 6: the fronts of ALL segments in a handful of launches ---------------------------------------------------------------
// annular_gram_all_f32: ONE gather of every segment's pixels into A_all [n][Ptot] (pix_all: flat pixel indices, the segments side
// by side, each padded with -1 = zero column to a whole number of K-slices of klen columns) and ONE ragged Gram product on the int8
// matrix cores (gram_i8_ragged_f32; seg_slice: device array of nseg + 1 slice offsets) -> G_all [nseg][n][n].
int annular_gram_all_f32(vipmi_ctx* ctx, const float* cube, int64_t n, int64_t P, const int32_t* pix_all, int64_t Ptot, int64_t klen,
                         const int32_t* seg_slice, int64_t nseg, float* A_all, double* G_all) {
  VIPMI_REQUIRE(seg_slice && A_all && G_all && (!cube || pix_all), "annular_gram_all: null pointer");
  VIPMI_REQUIRE(n > 0 && P > 0 && Ptot > 0 && nseg > 0, "annular_gram_all: bad sizes");
  if (cube) VIPMI_TRY(gather_f32(ctx, cube, n, P, pix_all, Ptot, A_all));       // (NULL: A_all already holds the matrix, e.g. scaled)
  return gram_i8_ragged_f32(ctx, A_all, n, Ptot, klen, seg_slice, nseg, G_all);
}

// annular_apply_all_f32: the coefficient matrices I - C of all segments (one launch, written transposed), their library windows
// (one launch) and ONE product over all segments that writes the residuals through the pixel list into cube_out
// (rowspace_scatter_kernel, project.hip).  lib_idx [nseg * n][m], lib_len [nseg * n], evals [nseg * n][m], evecs [nseg * n][m][m]
// as vipmi_annular_eigh_f64 leaves them; kseg: device array of the segments' numbers of components (<= kmax).
int annular_apply_all_f32(vipmi_ctx* ctx, const float* A_all, int64_t n, int64_t Ptot, const int32_t* tile_seg, const int32_t* pix_out,
                          int64_t nseg, const int32_t* lib_idx, const int32_t* lib_len, int64_t m, const double* G_all,
                          const double* evals, const double* evecs, const int32_t* kseg, int64_t kmax, int64_t P, float* cube_out,
                          const float* mu32) {
  VIPMI_REQUIRE(A_all && tile_seg && pix_out && lib_idx && lib_len && G_all && evals && evecs && kseg && cube_out,
                "annular_apply_all: null pointer");
  VIPMI_REQUIRE(n > 0 && Ptot > 0 && nseg > 0 && nseg <= 65535 && m > 0 && kmax > 0 && P > 0, "annular_apply_all: bad sizes");
  const int kld = (int)cdiv(n, 32) * 32, groups = (int)cdiv(n, 32);
  float* Wt = nullptr;
  int* frange = nullptr;
  VIPMI_TRY(ws(ctx, "ann_Wt_all", (size_t)nseg * n * kld, &Wt));
  if (ctx->opt("ann_range", 1) != 0) {
    VIPMI_TRY(ws(ctx, "ann_frange_all", (size_t)nseg * 2 * groups, &frange));
    hipLaunchKernelGGL(coeff_range_kernel, dim3(groups, (unsigned)nseg), dim3(1024), 0, ctx->stream, (int)n, lib_idx, lib_len, (int)m, frange);
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  VIPMI_CHECK_HIP(hipMemsetAsync(Wt, 0, sizeof(float) * (size_t)nseg * n * kld, ctx->stream));
  const size_t shm = (size_t)(m + kmax + 8) * sizeof(double);
  float* rho = nullptr;                  // mu32: A_all is D = M - 1 mu^T of a float64 cube; residuals = (I - C) D + rho mu^T
  if (mu32) VIPMI_TRY(ws(ctx, "ann_rho_all", (size_t)nseg * n, &rho));
  hipLaunchKernelGGL(coeff_kernel, dim3((unsigned)n, (unsigned)nseg), dim3(256), shm, ctx->stream, G_all, (int)n, lib_idx, lib_len, (int)m,
                     (int)m, evals, evecs, (int)kmax, (int)(Ptot < 2147483647 ? Ptot : 2147483647), Wt, rho, kseg, kld);
  VIPMI_CHECK_HIP(hipGetLastError());
  return rowspace_scatter_f32(ctx, Wt, kld, A_all, n, Ptot, tile_seg, pix_out, frange, P, cube_out, rho, mu32);
}

// ncomps: HOST array of nk truncation ranks (the reference's list `ncomp`, pca_local.py:665-668,892-902: one
// decomposition with max(ncomp), residuals for every V[:k]); residuals: [nk][n][npx].
int annular_residuals_multi_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* lib_idx,
                                const int32_t* lib_len, int64_t max_lib, const int32_t* ncomps, int64_t nk,
                                float* residuals) {
  VIPMI_REQUIRE(A && lib_idx && lib_len && residuals && ncomps, "annular_residuals: null pointer");
  VIPMI_REQUIRE(n > 0 && npx > 0 && max_lib > 0 && nk > 0, "annular_residuals: bad sizes");
  VIPMI_REQUIRE(max_lib <= n, "annular_residuals: max_lib > n");
  int64_t kmax = 0;
  for (int64_t i = 0; i < nk; ++i) {
    VIPMI_REQUIRE(ncomps[i] > 0, "annular_residuals: ncomp must be positive");
    if (ncomps[i] > kmax) kmax = ncomps[i];
  }
  const int m = (int)max_lib;
  double *G = nullptr, *H = nullptr, *evals = nullptr, *evecs = nullptr;
  VIPMI_TRY(ws(ctx, "ann_G", (size_t)n * n, &G));
  VIPMI_TRY(ws(ctx, "ann_H", (size_t)n * m * m, &H));
  VIPMI_TRY(ws(ctx, "ann_evals", (size_t)n * m, &evals));
  VIPMI_TRY(ws(ctx, "ann_evecs", (size_t)n * m * m, &evecs));
  VIPMI_TRY(annular_subgrams_f64(ctx, A, n, npx, lib_idx, lib_len, max_lib, m, G, H));
  if (m > ctx->opt("ann_large_min", 512)) {      // (option: A/B and test switch, at least 127)
    // libraries of more than 512 frames (max_frames_lib raised far above the reference's default 200) are beyond the
    // batched solver (one library per workgroup): the matrix-in-L2 solver takes them one after the other.  It has no
    // active-size argument and needs none: the sub-Gram matrices are zero padded, so the padding only adds zero
    // eigenvalues, which coeff_kernel skips, and the leading vectors are zero there.  (~5 ms per library of 600 frames.)
    VIPMI_REQUIRE(eigh_large_supported(m, kmax < m ? kmax : m), "annular_residuals: libraries of %d frames are not supported", m);
    VIPMI_TRY(eigh_large_f64(ctx, H, n, m, kmax < m ? kmax : m, evals, evecs));
  } else {
    VIPMI_TRY(eigh_leading(ctx, H, n, m, kmax < m ? kmax : m, lib_len, evals, evecs));
  }
  return annular_apply_f32(ctx, A, n, npx, lib_idx, lib_len, max_lib, m, G, evals, evecs, ncomps, nk, residuals);
}

int annular_residuals_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* lib_idx,
                          const int32_t* lib_len, int64_t max_lib, int64_t ncomp, float* residuals) {
  VIPMI_REQUIRE(ncomp > 0, "annular_residuals: bad sizes");
  const int32_t k = (int32_t)ncomp;
  return annular_residuals_multi_f32(ctx, A, n, npx, lib_idx, lib_len, max_lib, &k, 1, residuals);
}

}  // namespace vipmi
