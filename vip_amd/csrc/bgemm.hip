// bgemm.hip -- batched small products in "A B^T" form on the f32 matrix cores:
//     C[b] (M x N) = A0[ia[b]] (M x K) * B0[ib[b]]^T (N x K)   [ - A1[ia[b]] * B1[ib[b]]^T ]
// every operand row-major with the contraction index K contiguous.  It carries the FFT zoom of the ADI+mSDI path:
// the reference rescales every spectral channel of every frame with scale_fft (preproc/rescaling.py:1114-1217:
// zero-pad -> fft2 -> crop / pad the spectrum -> ifft2 -> real part -> crop / pad), which is a separable LINEAR map
//     Y = Re(E X E^T) = Er X Er^T - Ei X Ei^T
// with one small complex matrix E per scale factor (the reflect padding of cube_rescaling_wavelengths and the final
// crops fold into E as well).  With U = E X^T (an "A B^T" product) the result is Y = Er Ur^T - Ei Ui^T (another one),
// so a frame costs four real products of the frame size instead of two 2-D FFTs of awkward (non power-of-two) sizes.
//
// One wave owns a 64 x 64 (or 32 x 32) output tile of 16 x 16 v_mfma_f32_16x16x4_f32 blocks; operands are read straight from
// global memory in MFMA fragment layout (lane (r = lane & 15, kq = lane >> 4) reads row r, columns k0 + 4 kq .. + 3;
// component c feeds MFMA number c -- the same k permutation on both operands, so no LDS staging), as in gram.hip.
#include "common.h"

namespace vipmi {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool VEC>
__device__ __forceinline__ f32x4 ldfrag(const float* __restrict__ row, bool ok, int off, int K) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (!ok) return v;
  if (VEC && off + 4 <= K) {
    v = *reinterpret_cast<const f32x4*>(row + off);
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (off + c < K) v[c] = row[off + c];
  }
  return v;
}

struct BgemmArgs {
  const float* A0;
  const float* B0;
  const float* A1;      // may be null
  const float* B1;
  const int32_t* ia;    // per-batch index into the A arrays (null: b)
  const int32_t* ib;    // per-batch index into the B arrays (null: b)
  float* C;
  int M, N, K;
  int lda, ldb, ldc;
  int64_t sa, sb, sc;   // strides between matrices
};

template <bool VEC, int TB>       // TB x TB blocks of 16 x 16 per wave
__global__ __launch_bounds__(256) void bgemm_abt_kernel(BgemmArgs g, int tiles_n, int ntiles) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= ntiles) return;
  const int b = blockIdx.y;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int r = lane & 15, kq = lane >> 4;
  const int64_t oa = (int64_t)(g.ia ? g.ia[b] : b) * g.sa, ob = (int64_t)(g.ib ? g.ib[b] : b) * g.sb;
  const int nprod = g.A1 ? 2 : 1;
  f32x4 acc[TB][TB];
#pragma unroll
  for (int i = 0; i < TB; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  bool blka[TB], blkb[TB];            // wave-uniform: the block has rows inside the matrix (edge tiles skip the rest)
#pragma unroll
  for (int i = 0; i < TB; ++i) {
    blka[i] = (tm * TB + i) * 16 < g.M;
    blkb[i] = (tn * TB + i) * 16 < g.N;
  }
  for (int p = 0; p < nprod; ++p) {
    const float* A = (p ? g.A1 : g.A0) + oa;
    const float* B = (p ? g.B1 : g.B0) + ob;
    const float sign = p ? -1.f : 1.f;
    const float* pa[TB];
    const float* pb[TB];
    bool oka[TB], okb[TB];
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      const int ra = (tm * TB + i) * 16 + r, rb = (tn * TB + i) * 16 + r;
      oka[i] = ra < g.M;
      okb[i] = rb < g.N;
      pa[i] = A + (int64_t)(oka[i] ? ra : 0) * g.lda;
      pb[i] = B + (int64_t)(okb[i] ? rb : 0) * g.ldb;
    }
    // the fragments of step k0 + 16 are requested before the MFMAs of step k0 are issued
    f32x4 fa[TB], fb[TB];
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      fa[i] = ldfrag<VEC>(pa[i], oka[i], 4 * kq, g.K);
      fb[i] = ldfrag<VEC>(pb[i], okb[i], 4 * kq, g.K);
    }
    for (int k0 = 0; k0 < g.K; k0 += 16) {
      f32x4 na[TB], nb[TB];
#pragma unroll
      for (int i = 0; i < TB; ++i) {
        na[i] = ldfrag<VEC>(pa[i], oka[i] && k0 + 16 < g.K, k0 + 16 + 4 * kq, g.K);
        nb[i] = ldfrag<VEC>(pb[i], okb[i] && k0 + 16 < g.K, k0 + 16 + 4 * kq, g.K);
      }
#pragma unroll
      for (int i = 0; i < TB; ++i) fa[i] *= sign;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < TB; ++i) {
          if (!blka[i]) continue;
#pragma unroll
          for (int j = 0; j < TB; ++j) {
            if (!blkb[j]) continue;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][c], fb[j][c], acc[i][j], 0, 0, 0);
          }
        }
#pragma unroll
      for (int i = 0; i < TB; ++i) {
        fa[i] = na[i];
        fb[i] = nb[i];
      }
    }
  }
  float* C = g.C + (int64_t)b * g.sc;
  const int col = lane & 15;
#pragma unroll
  for (int i = 0; i < TB; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = (tm * TB + i) * 16 + (lane >> 4) * 4 + q;      // f32 16x16x4: row = (lane>>4)*4 + q
        const int cc = (tn * TB + j) * 16 + col;
        if (row < g.M && cc < g.N) C[(int64_t)row * g.ldc + cc] = acc[i][j][q];
      }
}

// Workgroup-tiled variant for matrices of at least 128 x 128: four waves share a 128 x 128 output tile (64 x 64 each);
// the two 128 x 16 operand panels of a k-step are staged through LDS once per workgroup (each panel feeds two waves),
// double buffered: the global loads of step k + 16 are in flight while step k is multiplied.  Halves the L2 -> L1
// traffic of the per-wave kernel above, which at 62 TF/s was bound by it (16 flop per operand byte).
constexpr int BG_LDP = 20;            // LDS row pitch of a panel in floats (80 bytes: conflict-free 16-byte fragment reads)

template <bool VEC>
__global__ __launch_bounds__(256) void bgemm_abt_lds_kernel(BgemmArgs g, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][128 * BG_LDP];      // [buffer][A | B][row][k]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int b = blockIdx.y;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int r = lane & 15, kq = lane >> 4;
  const int64_t oa = (int64_t)(g.ia ? g.ia[b] : b) * g.sa, ob = (int64_t)(g.ib ? g.ib[b] : b) * g.sb;
  const int nprod = g.A1 ? 2 : 1;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  bool blka[4], blkb[4];              // wave-uniform: block inside the matrix
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    blka[i] = tm * 128 + wm * 64 + i * 16 < g.M;
    blkb[i] = tn * 128 + wn * 64 + i * 16 < g.N;
  }
  // staging role of this thread: rows srow and srow + 64 of both panels, 4 consecutive k
  const int srow = threadIdx.x >> 2, sk = 4 * (threadIdx.x & 3);
  const int nk = (g.K + 15) / 16;
  const int nsteps = nk * nprod;
  auto panel_ptrs = [&](int step, const float*& A, const float*& B, float& sign, int& k0) {
    const int p = step / nk;
    A = (p ? g.A1 : g.A0) + oa;
    B = (p ? g.B1 : g.B0) + ob;
    sign = p ? -1.f : 1.f;
    k0 = (step % nk) * 16;
  };
  f32x4 ra[2], rb[2];
  auto fetch = [&](int step) {
    const float *A, *B;
    float sign;
    int k0;
    panel_ptrs(step, A, B, sign, k0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int rowa = tm * 128 + srow + 64 * h, rowb = tn * 128 + srow + 64 * h;
      ra[h] = ldfrag<VEC>(A + (int64_t)(rowa < g.M ? rowa : 0) * g.lda, rowa < g.M, k0 + sk, g.K) * sign;
      rb[h] = ldfrag<VEC>(B + (int64_t)(rowb < g.N ? rowb : 0) * g.ldb, rowb < g.N, k0 + sk, g.K);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<f32x4*>(&lds[buf][0][(srow + 64 * h) * BG_LDP + sk]) = ra[h];
      *reinterpret_cast<f32x4*>(&lds[buf][1][(srow + 64 * h) * BG_LDP + sk]) = rb[h];
    }
  };
  fetch(0);
  stash(0);
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int cur = step & 1;
    if (step + 1 < nsteps) fetch(step + 1);
    f32x4 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[i] = *reinterpret_cast<const f32x4*>(&lds[cur][0][(wm * 64 + i * 16 + r) * BG_LDP + 4 * kq]);
      fb[i] = *reinterpret_cast<const f32x4*>(&lds[cur][1][(wn * 64 + i * 16 + r) * BG_LDP + 4 * kq]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!blka[i]) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!blkb[j]) continue;
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][c], fb[j][c], acc[i][j], 0, 0, 0);
        }
      }
    if (step + 1 < nsteps) stash(cur ^ 1);
    __syncthreads();
  }
  float* C = g.C + (int64_t)b * g.sc;
  const int col = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = tm * 128 + wm * 64 + i * 16 + (lane >> 4) * 4 + q;      // f32 16x16x4: row = (lane>>4)*4 + q
        const int cc = tn * 128 + wn * 64 + j * 16 + col;
        if (row < g.M && cc < g.N) C[(int64_t)row * g.ldc + cc] = acc[i][j][q];
      }
}

}  // namespace

int bgemm_abt_f32(vipmi_ctx* ctx, const float* A0, const float* B0, const float* A1, const float* B1,
                  const int32_t* ia, const int32_t* ib, int64_t nbatch, int64_t M, int64_t N, int64_t K, int64_t lda,
                  int64_t ldb, int64_t ldc, int64_t sa, int64_t sb, int64_t sc, float* C) {
  VIPMI_REQUIRE(A0 && B0 && C, "bgemm: null pointer");
  VIPMI_REQUIRE((A1 == nullptr) == (B1 == nullptr), "bgemm: second product needs both operands");
  VIPMI_REQUIRE(nbatch > 0 && M > 0 && N > 0 && K > 0 && lda >= K && ldb >= K && ldc >= N, "bgemm: bad sizes");
  VIPMI_REQUIRE(nbatch <= 65535, "bgemm: more than 65535 matrices per call");
  StageScope scope(ctx, "bgemm");
  BgemmArgs g{A0, B0, A1, B1, ia, ib, C, (int)M, (int)N, (int)K, (int)lda, (int)ldb, (int)ldc, sa, sb, sc};
  // 64 x 64 tiles per wave (64 MFMAs per 8 fragment loads) once a matrix has enough of them, else 32 x 32
  const int tb = (M >= 128 && N >= 128 && ctx->opt("bgemm_tb", 4) == 4) ? 4 : 2;
  const int tiles_m = (int)cdiv(M, 16 * tb), tiles_n = (int)cdiv(N, 16 * tb), ntiles = tiles_m * tiles_n;
  auto aligned = [](const float* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec = lda % 4 == 0 && ldb % 4 == 0 && sa % 4 == 0 && sb % 4 == 0 && aligned(A0) && aligned(B0) &&
                   aligned(A1) && aligned(B1);
  if (M >= 128 && N >= 128 && ctx->opt("bgemm_lds", 1) != 0) {
    const int tm = (int)cdiv(M, 128), tn = (int)cdiv(N, 128);
    dim3 grid_l((unsigned)(tm * tn), (unsigned)nbatch);
    if (vec)
      hipLaunchKernelGGL(bgemm_abt_lds_kernel<true>, grid_l, dim3(256), 0, ctx->stream, g, tn);
    else
      hipLaunchKernelGGL(bgemm_abt_lds_kernel<false>, grid_l, dim3(256), 0, ctx->stream, g, tn);
    VIPMI_CHECK_HIP(hipGetLastError());
    return VIPMI_OK;
  }
  dim3 grid((unsigned)cdiv(ntiles, 4), (unsigned)nbatch), block(256);
  if (vec && tb == 4)
    hipLaunchKernelGGL((bgemm_abt_kernel<true, 4>), grid, block, 0, ctx->stream, g, tiles_n, ntiles);
  else if (vec)
    hipLaunchKernelGGL((bgemm_abt_kernel<true, 2>), grid, block, 0, ctx->stream, g, tiles_n, ntiles);
  else if (tb == 4)
    hipLaunchKernelGGL((bgemm_abt_kernel<false, 4>), grid, block, 0, ctx->stream, g, tiles_n, ntiles);
  else
    hipLaunchKernelGGL((bgemm_abt_kernel<false, 2>), grid, block, 0, ctx->stream, g, tiles_n, ntiles);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

}  // namespace vipmi
