// derotate_fft.hip -- FFT fast path of the reference's 3-shear rotation (preproc/derotation.py:
// 542-640) for power-of-two padded lengths Le in {512, 1024, 2048, 4096} (frames of 128/256/512/1024 px).
//
// One circular sinc shift  y = ifft( fft(x) * exp(-2 pi i f s) )  of an Le-point complex line is done by
// ONE WAVE entirely on chip:
//   * the line lives in registers, Le/64 complex values per lane;
//   * Le = R1*R2*R3 (radices 8/16): forward transform = decimation-in-frequency (butterfly, twiddle,
//     exchange), which leaves the spectrum digit-reversed across lanes/registers; the shear phase is
//     applied right there (every lane knows the frequency index of each of its registers) and the
//     inverse transform = decimation-in-time walking the same stages backwards -- so no reordering pass
//     is ever needed and a line costs 4 lane<->lane exchanges through a wave-private LDS region
//     (padded layouts: all ds_read_b64 / ds_write_b64 patterns are bank-conflict free);
//   * inter-stage twiddles are per-lane constants: computed once per wave (float64 sincospi) and kept
//     in registers while the wave loops over many lines;
//   * the shear phase exp(-2 pi i k s/Le) factorises over the digits of k: U3 + R3 sincos per line
//     (argument reduced in float64), the rest are complex multiplies.
// Three kernels per frame batch:  rows (real frame -> complex A1), columns (A1 -> A2, only the N rows
// that survive the final crop are produced), rows (A2 -> real output, crop + NaN/zero mask restore).
// The column kernel moves [N x 8 columns] tiles through LDS so that global traffic is 64-byte row
// segments instead of 8-byte strided accesses.  Zero structure exploited: shear 1 processes N of the
// Le rows, shear 2 writes N of Le rows, shear 3 processes N rows.
#include "common.h"
#include "rot_common.h"
#include "fft_wave.h"

namespace vipmi {

namespace {

using namespace fftw;

// ---- shear 1: rows; real input gathered from the frame through the rot90 index map ----
template <class P>
__global__ __launch_bounds__(64 * P::WPB) void fft_shear1(const float* __restrict__ in,
                                                          const RotFrame* __restrict__ fr, RotGeom g,
                                                          cf* __restrict__ A1, int f0, int nf,
                                                          const cf* __restrict__ twtab) {
  extern __shared__ __attribute__((aligned(16))) cf lds_all[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform
  const int sub = wave % P::WPL, slot = wave / P::WPL;
  cf* lds = lds_all + slot * P::LDS_ELEMS;
  Twiddles<P> tw;
  tw.init(twtab, lds_all + P::LPB * P::LDS_ELEMS, lane);   // table sits after the line regions
  const int nlines = nf * g.N;
  const int niter = (nlines + gridDim.x * P::LPB - 1) / (gridDim.x * P::LPB);   // uniform trip count
  for (int it = 0; it < niter; ++it) {
    int line = (it * gridDim.x + blockIdx.x) * P::LPB + slot;
    const bool live = line < nlines;
    if (!live) line = nlines - 1;               // idle slots redo the last line (keeps barriers uniform)
    const int fl = line / g.N, yrel = line % g.N, f = f0 + fl;
    const RotFrame p = fr[f];
    const int r0 = (p.q == 1 || p.q == 2) ? g.alt0 : g.off;
    const int c0 = (p.q == 2 || p.q == 3) ? g.alt0 : g.off;
    const int Y = r0 + yrel;
    const float* frame = in + (int64_t)f * g.N * g.N;
    // source pixel of canvas'(Y, X) is frame[base + X*stride] (rot90 folded into an affine index map)
    int base, stride;
    switch (p.q) {
      case 1: stride = g.N; base = -g.off * g.N + (g.Lc - Y - g.off); break;
      case 2: stride = -1; base = (g.Lc - Y - g.off) * g.N + (g.Lc - g.off); break;
      case 3: stride = -g.N; base = (g.Lc - g.off) * g.N + (Y - g.off); break;
      default: stride = 1; base = (Y - g.off) * g.N - g.off; break;
    }
    cf v[P::VL];
#pragma unroll
    for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
      for (int n1 = 0; n1 < P::R1; ++n1) {
        float val = 0.f;
        if (n1 >= P::NLO && n1 <= P::NLO + P::NCNT) {       // compile-time window
          const int X = P::M1 * n1 + lane + 64 * (sub * P::U1L + ul);
          if (X >= c0 && X < c0 + g.N) {
            const float t = frame[base + X * stride];
            val = (t == t) ? t : 0.f;
          }
        }
        v[ul * P::R1 + n1] = mkcf(val, 0.f);
      }
    line_shift<P>(v, tw, lds, p.a * (double)(Y - g.c), lane, sub);
    if (live) {
      cf* orow = A1 + ((int64_t)fl * g.N + yrel) * P::L;
#pragma unroll
      for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
        for (int n1 = 0; n1 < P::R1; ++n1)
          orow[P::M1 * n1 + lane + 64 * (sub * P::U1L + ul)] = v[ul * P::R1 + n1];
    }
  }
}

// ---- shear 2: columns, LPB adjacent columns per workgroup, tiles staged through LDS ----
template <class P>
__global__ __launch_bounds__(64 * P::WPB) void fft_shear2(const cf* __restrict__ A1,
                                                          const RotFrame* __restrict__ fr, RotGeom g,
                                                          cf* __restrict__ A2, int f0, int nf,
                                                          const cf* __restrict__ twtab) {
  extern __shared__ __attribute__((aligned(16))) cf lds_all[];
  constexpr int W = P::LPB, LDT = W + 1;      // tile row stride (complex): conflict-free column reads
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform
  const int sub = wave % P::WPL, slot = wave / P::WPL;
  cf* lds = lds_all + slot * P::LDS_ELEMS;
  cf* tile = lds_all;                          // [N][LDT], aliases the exchange regions between phases
  Twiddles<P> tw;
  tw.init(twtab, lds_all + P::LPB * P::LDS_ELEMS, lane);   // table sits after the line regions
  const int groups = P::L / W;
  const int units = nf * groups;
  for (int uu = blockIdx.x * 2; uu < units; uu += gridDim.x * 2) {
    for (int h = 0; h < 2; ++h) {
      const int unit = uu + h;
      if (unit >= units) break;                // uniform across the workgroup
      const int fl = unit / groups, X0 = (unit % groups) * W, f = f0 + fl;
      const RotFrame p = fr[f];
      const int r0 = (p.q == 1 || p.q == 2) ? g.alt0 : g.off;
      const cf* src = A1 + (int64_t)fl * g.N * P::L + X0;
      for (int e = threadIdx.x; e < g.N * W; e += 64 * P::WPB) {
        const int row = e / W, c = e % W;
        tile[row * LDT + c] = src[(int64_t)row * P::L + c];
      }
      __syncthreads();
      cf v[P::VL];
#pragma unroll
      for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
        for (int n1 = 0; n1 < P::R1; ++n1) {
          cf val = mkcf(0.f, 0.f);
          if (n1 >= P::NLO && n1 <= P::NLO + P::NCNT) {     // compile-time window
            const int yrel = P::M1 * n1 + lane + 64 * (sub * P::U1L + ul) - r0;
            if (yrel >= 0 && yrel < g.N) val = tile[yrel * LDT + slot];
          }
          v[ul * P::R1 + n1] = val;
        }
      __syncthreads();
      const int X = X0 + slot;
        line_shift<P>(v, tw, lds, p.b * (double)(X - g.c), lane, sub);
      __syncthreads();
#pragma unroll
      for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
        for (int n1 = P::NLO; n1 < P::NLO + P::NCNT; ++n1) {
          const int m = P::M1 * (n1 - P::NLO) + lane + 64 * (sub * P::U1L + ul);      // off == M1*NLO
          tile[m * LDT + slot] = v[ul * P::R1 + n1];
        }
      __syncthreads();
      cf* dst = A2 + (int64_t)fl * g.N * P::L + X0;
      for (int e = threadIdx.x; e < g.N * W; e += 64 * P::WPB) {
        const int row = e / W, c = e % W;
        dst[(int64_t)row * P::L + c] = tile[row * LDT + c];
      }
      __syncthreads();
    }
  }
}

// ---- shear 3: rows of A2 -> real part, crop, mask restore ----
template <class P>
__global__ __launch_bounds__(64 * P::WPB) void fft_shear3(const cf* __restrict__ A2,
                                                          const RotFrame* __restrict__ fr, RotGeom g,
                                                          const float* __restrict__ in,
                                                          float* __restrict__ out, int f0, int nf,
                                                          int mask_nan, int mask_zero,
                                                          const cf* __restrict__ twtab) {
  extern __shared__ __attribute__((aligned(16))) cf lds_all[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform
  const int sub = wave % P::WPL, slot = wave / P::WPL;
  cf* lds = lds_all + slot * P::LDS_ELEMS;
  Twiddles<P> tw;
  tw.init(twtab, lds_all + P::LPB * P::LDS_ELEMS, lane);   // table sits after the line regions
  const int nlines = nf * g.N;
  const int niter = (nlines + gridDim.x * P::LPB - 1) / (gridDim.x * P::LPB);   // uniform trip count
  for (int it = 0; it < niter; ++it) {
    int line = (it * gridDim.x + blockIdx.x) * P::LPB + slot;
    const bool live = line < nlines;
    if (!live) line = nlines - 1;
    const int fl = line / g.N, m = line % g.N, f = f0 + fl;
    const RotFrame p = fr[f];
    const int Y = g.off + m;
    const cf* irow = A2 + ((int64_t)fl * g.N + m) * P::L;
    cf v[P::VL];
#pragma unroll
    for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
      for (int n1 = 0; n1 < P::R1; ++n1) v[ul * P::R1 + n1] = irow[P::M1 * n1 + lane + 64 * (sub * P::U1L + ul)];
    line_shift<P>(v, tw, lds, p.a * (double)(Y - g.c), lane, sub);
    if (live) {
      const int64_t obase = ((int64_t)f * g.N + m) * g.N;
#pragma unroll
      for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
        for (int n1 = P::NLO; n1 < P::NLO + P::NCNT; ++n1) {
          const int j = P::M1 * (n1 - P::NLO) + lane + 64 * (sub * P::U1L + ul);        // off == M1*NLO
          float re = v[ul * P::R1 + n1].x;
          const float src = in[obase + j];
          if (mask_nan && !(src == src)) re = __uint_as_float(0x7fc00000u);
          if (mask_zero && src == 0.f) re = 0.f;
          out[obase + j] = re;
        }
    }
  }
}

template <class P>
int run_plan(vipmi_ctx* ctx, const float* in, const RotFrame* d_frames, const RotGeom& g, int64_t n, float* out,
             int mask_nan, int mask_zero) {
  const int64_t per_frame = (int64_t)g.N * P::L;
  int64_t chunk = ctx->opt("rot_batch", 0);
  if (chunk <= 0) {
    int64_t budget = ctx->opt("rot_ws_mb", 4096) * (int64_t)(1 << 20);
    chunk = budget / (2 * per_frame * (int64_t)sizeof(cf));
  }
  if (chunk < 1) chunk = 1;
  if (chunk > n) chunk = n;
  cf *A1 = nullptr, *A2 = nullptr;
  VIPMI_TRY(ws(ctx, "rot_a1", (size_t)(chunk * per_frame), &A1));
  VIPMI_TRY(ws(ctx, "rot_a2", (size_t)(chunk * per_frame), &A2));
  size_t lds = (size_t)P::LPB * P::LDS_ELEMS * sizeof(cf);
  const size_t tile = (size_t)g.N * (P::LPB + 1) * sizeof(cf);
  VIPMI_REQUIRE(tile <= lds, "derotate(fft): staging tile larger than the exchange regions");
  lds += (size_t)Twiddles<P>::LDS_ELEMS * sizeof(cf);
  VIPMI_REQUIRE(lds <= 160 * 1024, "derotate(fft): LDS budget exceeded (%zu)", lds);
  cf* twtab = nullptr;
  {
    std::vector<cf> tab;
    Twiddles<P>::fill_table(tab);
    char key[32];
    snprintf(key, sizeof key, "L%d", P::L);
    void* pt = nullptr;
    VIPMI_TRY(ctx->upload_cached("rot_twiddles", key, tab.data(), tab.size() * sizeof(cf), &pt));
    twtab = reinterpret_cast<cf*>(pt);
  }
  auto k1 = fft_shear1<P>;
  auto k2 = fft_shear2<P>;
  auto k3 = fft_shear3<P>;
  VIPMI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  VIPMI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  VIPMI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int wgs_per_cu = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;
  // "reserve_cus": leave some CUs to a concurrently running latency-bound kernel of another stream
  // (the 13-workgroup Jacobi eigensolver of the next cube when calls are pipelined over two streams)
  int usable_cu = ctx->num_cu - (int)ctx->opt("reserve_cus", 0);
  if (usable_cu < 1) usable_cu = 1;
  const int maxwg = usable_cu * wgs_per_cu;
  for (int64_t f0 = 0; f0 < n; f0 += chunk) {
    const int nf = (int)((n - f0) < chunk ? (n - f0) : chunk);
    const int64_t nlines = (int64_t)nf * g.N;
    int gr = (int)cdiv(nlines, P::LPB);
    if (gr > maxwg) gr = maxwg;
    const int64_t units = (int64_t)nf * (P::L / P::LPB);
    int gc = (int)cdiv(units, 2);
    if (gc > maxwg) gc = maxwg;
    ctx->tic("k_rot_s1");
    hipLaunchKernelGGL(k1, dim3(gr), dim3(64 * P::WPB), lds, ctx->stream, in, d_frames, g, A1, (int)f0, nf, twtab);
    ctx->toc("k_rot_s1");
    ctx->tic("k_rot_s2");
    hipLaunchKernelGGL(k2, dim3(gc), dim3(64 * P::WPB), lds, ctx->stream, A1, d_frames, g, A2, (int)f0, nf, twtab);
    ctx->toc("k_rot_s2");
    ctx->tic("k_rot_s3");
    hipLaunchKernelGGL(k3, dim3(gr), dim3(64 * P::WPB), lds, ctx->stream, A2, d_frames, g, in, out, (int)f0, nf,
                       mask_nan, mask_zero, twtab);
    ctx->toc("k_rot_s3");
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  return VIPMI_OK;
}

}  // namespace

bool derotate_fft_supported(const RotGeom& g) {
  // power-of-two frames: N = Le/4 centred at 3Le/8 (the kernels rely on this alignment)
  return (g.Le == 512 || g.Le == 1024 || g.Le == 2048 || g.Le == 4096) && g.L == g.Le &&
         g.N * 4 == g.Le && g.off * 8 == 3 * g.Le;
}

int derotate_fft(vipmi_ctx* ctx, const float* in, const RotFrame* d_frames, const RotGeom& g, int64_t n,
                 float* out, int mask_nan, int mask_zero) {
  switch (g.Le) {
    case 512: return run_plan<Plan512>(ctx, in, d_frames, g, n, out, mask_nan, mask_zero);
    case 1024: return run_plan<Plan1024>(ctx, in, d_frames, g, n, out, mask_nan, mask_zero);
    case 2048: return run_plan<Plan2048>(ctx, in, d_frames, g, n, out, mask_nan, mask_zero);
    case 4096: return run_plan<Plan4096>(ctx, in, d_frames, g, n, out, mask_nan, mask_zero);
    default:
      set_error("derotate(fft): unsupported padded length %d", g.Le);
      return VIPMI_ERR_UNSUPPORTED;
  }
}

}  // namespace vipmi
