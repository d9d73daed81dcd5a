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

namespace vipmi {

namespace {

typedef float2 cf;

__device__ __forceinline__ cf cmul(cf a, cf b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cf cmulc(cf a, cf b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)
__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward) or +i (inverse)
template <bool INV>
__device__ __forceinline__ cf mul_mi(cf a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }

// w16^e = exp(-2 pi i e/16), e = 0..7
__device__ __forceinline__ cf w16(int e) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R = 0.70710678118654752f;
  switch (e & 7) {
    case 0: return make_float2(1.f, 0.f);
    case 1: return make_float2(C1, -S1);
    case 2: return make_float2(R, -R);
    case 3: return make_float2(S1, -C1);
    case 4: return make_float2(0.f, -1.f);
    case 5: return make_float2(-S1, -C1);
    case 6: return make_float2(-R, -R);
    default: return make_float2(-C1, -S1);
  }
}

template <bool INV>
__device__ __forceinline__ cf twc(cf v, int e16) {   // v * w16^e (forward) or v * conj(w16^e) (inverse)
  const cf w = w16(e16);
  return INV ? cmulc(v, w) : cmul(v, w);
}

template <bool INV>
__device__ __forceinline__ void dft4(cf& a0, cf& a1, cf& a2, cf& a3) {
  const cf t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_mi<INV>(csub(a1, a3));
  a0 = cadd(t0, t2);
  a2 = csub(t0, t2);
  a1 = cadd(t1, t3);
  a3 = csub(t1, t3);
}

// natural-order in-place small DFTs, stride-1 arrays of R complex registers
template <int R, bool INV>
struct Dft;

template <bool INV>
struct Dft<4, INV> {
  static __device__ __forceinline__ void run(cf* v) { dft4<INV>(v[0], v[1], v[2], v[3]); }
};

template <bool INV>
struct Dft<8, INV> {
  static __device__ __forceinline__ void run(cf* v) {
    // n = 2*n1 + n2: two 4-point DFTs over n1, twiddle w8^(n2*k1), 2-point DFTs over n2
    cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4<INV>(e0, e1, e2, e3);
    dft4<INV>(o0, o1, o2, o3);
    o1 = twc<INV>(o1, 2);
    o2 = mul_mi<INV>(o2);
    o3 = twc<INV>(o3, 6);
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
  }
};

template <bool INV>
struct Dft<16, INV> {
  static __device__ __forceinline__ void run(cf* v) {
    // n = 4*n1 + n2, k = k1 + 4*k2: DFT4 over n1 -> twiddle w16^(n2*k1) -> DFT4 over n2
    cf y[4][4];   // y[n2][k1]
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
      y[n2][0] = v[n2]; y[n2][1] = v[4 + n2]; y[n2][2] = v[8 + n2]; y[n2][3] = v[12 + n2];
      dft4<INV>(y[n2][0], y[n2][1], y[n2][2], y[n2][3]);
    }
#pragma unroll
    for (int n2 = 1; n2 < 4; ++n2)
#pragma unroll
      for (int k1 = 1; k1 < 4; ++k1) {
        const int e = n2 * k1;                     // 1,2,3,2,4,6,3,6,9
        if (e == 4) y[n2][k1] = mul_mi<INV>(y[n2][k1]);
        else if (e == 9) { cf t = twc<INV>(y[n2][k1], 1); y[n2][k1] = make_float2(-t.x, -t.y); }  // w16^9 = -w16^1
        else y[n2][k1] = twc<INV>(y[n2][k1], e);
      }
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
      dft4<INV>(y[0][k1], y[1][k1], y[2][k1], y[3][k1]);
      v[k1] = y[0][k1]; v[k1 + 4] = y[1][k1]; v[k1 + 8] = y[2][k1]; v[k1 + 12] = y[3][k1];
    }
  }
};

template <int R1_, int R2_, int R3_, int S1_, int T1_, int T2_, int WPB_, int WPL_>
struct Plan {
  static constexpr int R1 = R1_, R2 = R2_, R3 = R3_;
  static constexpr int L = R1 * R2 * R3, M1 = L / R1, M2 = R3;
  static constexpr int U1 = M1 / 64, U2 = R1 * M2 / 64, U3 = R1 * R2 / 64;
  static constexpr int VPT = L / 64;
  static constexpr int S1 = S1_, T1 = T1_, T2 = T2_;   // LDS strides (complex elements)
  static constexpr int LDS_ELEMS = (R1 * S1 > R1 * T1) ? R1 * S1 : R1 * T1;
  static constexpr int WPB = WPB_;                     // waves per workgroup
  static constexpr int WPL = WPL_;                     // waves cooperating on one line
  static constexpr int LPB = WPB_ / WPL_;              // lines in flight per workgroup
  static constexpr int U1L = U1 / WPL_, U2L = U2 / WPL_, U3L = U3 / WPL_;
  static constexpr int VL = VPT / WPL_;                // complex registers per lane
  static_assert(U1L * WPL_ == U1 && U2L * WPL_ == U2 && U3L * WPL_ == U3, "radix plan not divisible by WPL");
  // a frame of N = L/4 pixels sits at canvas offset 3L/8: line elements M1*n1 + n2 with
  // n1 in [NLO, NLO + NCNT) are exactly the N output positions; inputs may be shifted by one
  // (rot90 pre-step), which adds n1 = NLO + NCNT.
  static constexpr int NLO = 3 * R1_ / 8, NCNT = R1_ / 4;
  static_assert(U1 * R1 == VPT && U2 * R2 == VPT && U3 * R3 == VPT, "bad radix plan");
};
// WPL > 1: the line is split over WPL waves (VL = 16 complex registers per lane instead of 32/64), which
// keeps the kernels under 128 VGPRs (4 waves/SIMD, no spills); the exchanges then need a workgroup barrier.
using Plan512 = Plan<8, 8, 8, 72, 72, 9, 8, 1>;
using Plan1024 = Plan<16, 8, 8, 72, 72, 9, 8, 1>;
using Plan2048 = Plan<16, 16, 8, 136, 152, 9, 16, 2>;
using Plan4096 = Plan<16, 16, 16, 272, 272, 17, 16, 4>;

// exp(-2 pi i j/64), j = 0..63
__device__ const float ROOT64_C[64] = {1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f, 6.123233996e-17f, -9.801714033e-02f, -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f, -4.713967368e-01f, -5.555702330e-01f, -6.343932842e-01f, -7.071067812e-01f, -7.730104534e-01f, -8.314696123e-01f, -8.819212643e-01f, -9.238795325e-01f, -9.569403357e-01f, -9.807852804e-01f, -9.951847267e-01f, -1.000000000e+00f, -9.951847267e-01f, -9.807852804e-01f, -9.569403357e-01f, -9.238795325e-01f, -8.819212643e-01f, -8.314696123e-01f, -7.730104534e-01f, -7.071067812e-01f, -6.343932842e-01f, -5.555702330e-01f, -4.713967368e-01f, -3.826834324e-01f, -2.902846773e-01f, -1.950903220e-01f, -9.801714033e-02f, -1.836970199e-16f, 9.801714033e-02f, 1.950903220e-01f, 2.902846773e-01f, 3.826834324e-01f, 4.713967368e-01f, 5.555702330e-01f, 6.343932842e-01f, 7.071067812e-01f, 7.730104534e-01f, 8.314696123e-01f, 8.819212643e-01f, 9.238795325e-01f, 9.569403357e-01f, 9.807852804e-01f, 9.951847267e-01f};
__device__ const float ROOT64_S[64] = {-0.000000000e+00f, -9.801714033e-02f, -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f, -4.713967368e-01f, -5.555702330e-01f, -6.343932842e-01f, -7.071067812e-01f, -7.730104534e-01f, -8.314696123e-01f, -8.819212643e-01f, -9.238795325e-01f, -9.569403357e-01f, -9.807852804e-01f, -9.951847267e-01f, -1.000000000e+00f, -9.951847267e-01f, -9.807852804e-01f, -9.569403357e-01f, -9.238795325e-01f, -8.819212643e-01f, -8.314696123e-01f, -7.730104534e-01f, -7.071067812e-01f, -6.343932842e-01f, -5.555702330e-01f, -4.713967368e-01f, -3.826834324e-01f, -2.902846773e-01f, -1.950903220e-01f, -9.801714033e-02f, -1.224646799e-16f, 9.801714033e-02f, 1.950903220e-01f, 2.902846773e-01f, 3.826834324e-01f, 4.713967368e-01f, 5.555702330e-01f, 6.343932842e-01f, 7.071067812e-01f, 7.730104534e-01f, 8.314696123e-01f, 8.819212643e-01f, 9.238795325e-01f, 9.569403357e-01f, 9.807852804e-01f, 9.951847267e-01f, 1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f};
__device__ __forceinline__ cf root64(int j) { return make_float2(ROOT64_C[j & 63], ROOT64_S[j & 63]); }

template <class P>
struct Twiddles {
  // Stage-1 twiddle w_L^(n2*k1), n2 = lane + 64u, k1 = 4h + l, factorises into
  //   w_L^(lane*l) * w_L^(4*lane*h)        (per lane: 3 + (R1/4 - 1) complex)
  //   * w_(L/64)^(u*k1) = root64(...)       (lane independent, u > 0 only)
  // and the stage-2 twiddle w_M1^(b*ka), b = lane % M2, ka = 4h + l, likewise.  The per-lane digit
  // factors (PER_LANE complex, float64-accurate, computed on the host) live in a small LDS table
  // [PER_LANE][64 lanes] shared by all waves of the workgroup and are re-read where they are used:
  // keeping them (or, worse, all R-1 products, which LICM would otherwise rebuild) in registers pushes
  // the Le = 2048 kernels over 128 VGPRs and makes them spill inside the line loop.
  static constexpr int N1H = P::R1 / 4 - 1, N2H = P::R2 / 4 - 1;
  static constexpr int PER_LANE = 6 + N1H + N2H;          // table entries per lane
  static constexpr int LDS_ELEMS = PER_LANE * 64;
  const cf* tab;                                          // LDS, already offset by lane

  __device__ __forceinline__ void init(const cf* __restrict__ gtab, cf* __restrict__ ltab, int lane) {
    for (int e = threadIdx.x; e < PER_LANE * 64; e += blockDim.x) {
      const int ln = e / PER_LANE, j = e % PER_LANE;
      ltab[j * 64 + ln] = gtab[e];
    }
    __syncthreads();
    tab = ltab + lane;
  }
  static void fill_table(std::vector<cf>& tabv) {
    tabv.resize(64 * PER_LANE);
    auto unit = [](long e, long period) {
      const double ang = -2.0 * M_PI * (double)(e % period) / (double)period;
      return make_float2((float)cos(ang), (float)sin(ang));
    };
    for (int lane = 0; lane < 64; ++lane) {
      cf* t = &tabv[lane * PER_LANE];
      const int b = lane % P::M2;
      for (int l = 1; l < 4; ++l) t[l - 1] = unit(lane * l, P::L);
      for (int h = 1; h <= N1H; ++h) t[3 + h - 1] = unit(4 * lane * h, P::L);
      for (int l = 1; l < 4; ++l) t[3 + N1H + l - 1] = unit(b * l, P::M1);
      for (int h = 1; h <= N2H; ++h) t[6 + N1H + h - 1] = unit(4 * b * h, P::M1);
    }
  }
  __device__ __forceinline__ cf t1l(int l) const { return tab[(l - 1) * 64]; }
  __device__ __forceinline__ cf t1h(int h) const { return tab[(3 + h - 1) * 64]; }
  __device__ __forceinline__ cf t2l(int l) const { return tab[(3 + N1H + l - 1) * 64]; }
  __device__ __forceinline__ cf t2h(int h) const { return tab[(6 + N1H + h - 1) * 64]; }
  template <bool CONJ>
  __device__ __forceinline__ cf apply1(cf v, int u, int k1) const {
    const int h = k1 >> 2, l = k1 & 3;
    cf w;
    if (h && l) w = cmul(t1h(h), t1l(l));
    else if (h) w = t1h(h);
    else w = t1l(l);
    if (u != 0) w = cmul(w, root64(u * k1 * (4096 / P::L)));   // u is wave-uniform
    return CONJ ? cmulc(v, w) : cmul(v, w);
  }
  template <bool CONJ>
  __device__ __forceinline__ cf apply2(cf v, int ka) const {
    const int h = ka >> 2, l = ka & 3;
    cf w;
    if (h && l) w = cmul(t2h(h), t2l(l));
    else if (h) w = t2h(h);
    else w = t2l(l);
    return CONJ ? cmulc(v, w) : cmul(v, w);
  }
};

template <class P>
__device__ __forceinline__ void xbar() {
  if (P::WPL > 1) __syncthreads();          // all waves of the workgroup run the same line count
  else __builtin_amdgcn_wave_barrier();
}

// y = ifft(fft(x) * exp(-2 pi i f s)) for a line distributed over WPL waves (sub = wave % WPL):
// distribution D1 in and out: v[ul*R1 + n1] = x[M1*n1 + lane + 64*(sub*U1L + ul)].
template <class P>
__device__ __forceinline__ void line_shift(cf (&v)[P::VL], const Twiddles<P>& tw, cf* __restrict__ lds,
                                           double s, int lane, int sub) {
  constexpr int R1 = P::R1, R2 = P::R2, R3 = P::R3, M2 = P::M2;
  // ---------------- forward (DIF) ----------------
#pragma unroll
  for (int ul = 0; ul < P::U1L; ++ul) {
    const int u = sub * P::U1L + ul;
    Dft<R1, false>::run(&v[ul * R1]);
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1) v[ul * R1 + k1] = tw.template apply1<false>(v[ul * R1 + k1], u, k1);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) lds[k1 * P::S1 + lane + 64 * u] = v[ul * R1 + k1];
  }
  xbar<P>();
#pragma unroll
  for (int ul = 0; ul < P::U2L; ++ul) {
    const int t = lane + 64 * (sub * P::U2L + ul), k1 = t / M2, b = t % M2;
#pragma unroll
    for (int a = 0; a < R2; ++a) v[ul * R2 + a] = lds[k1 * P::S1 + M2 * a + b];
  }
  xbar<P>();
#pragma unroll
  for (int ul = 0; ul < P::U2L; ++ul) {
    const int t = lane + 64 * (sub * P::U2L + ul), k1 = t / M2, b = t % M2;
    Dft<R2, false>::run(&v[ul * R2]);
#pragma unroll
    for (int ka = 1; ka < R2; ++ka) v[ul * R2 + ka] = tw.template apply2<false>(v[ul * R2 + ka], ka);
#pragma unroll
    for (int ka = 0; ka < R2; ++ka) lds[k1 * P::T1 + ka * P::T2 + b] = v[ul * R2 + ka];
  }
  xbar<P>();
#pragma unroll
  for (int ul = 0; ul < P::U3L; ++ul) {
    const int w = lane + 64 * (sub * P::U3L + ul), k1 = w / R2, ka = w % R2;
#pragma unroll
    for (int b = 0; b < R3; ++b) v[ul * R3 + b] = lds[k1 * P::T1 + ka * P::T2 + b];
  }
  xbar<P>();
  // ---------------- spectrum: phase ramp of the shear, 1/L normalisation ----------------
  // frequency index of v[ul*R3 + kb] is k = k1 + R1*ka + R1*R2*kb with (k1, ka) = divmod(lane + 64u, R2),
  // signed (numpy fftfreq order) through kb.  exp(-2 pi i k s/L) = pa0 * z^u * w^kbs with
  //   pa0 = exp(-2 pi i (lane/R2 + R1*(lane%R2)) s/L)   (per lane)
  //   z   = exp(-2 pi i (64/R2) s/L),  w = exp(-2 pi i R1 R2 s/L)   (wave-uniform)
  // a few sincos per line (arguments reduced in float64), the rest are complex multiplies.
  {
    const double sl = s / (double)P::L;
    auto expi = [](double turns) {
      turns -= rint(turns);
      float sn, cs;
      sincospif((float)(-2.0 * turns), &sn, &cs);
      return make_float2(cs, sn);
    };
    const cf w = expi((double)(R1 * R2) * sl);
    const cf z = expi((double)(64 / R2) * sl);
    const int u0 = sub * P::U3L;
    cf pa = expi((double)(lane / R2 + (64 / R2) * u0 + R1 * (lane % R2)) * sl);
    pa = make_float2(pa.x * (1.0f / (float)P::L), pa.y * (1.0f / (float)P::L));
    cf pb[R3];                                   // w^kbs, kbs = 0..R3/2-1, -R3/2..-1
    pb[0] = make_float2(1.f, 0.f);
    pb[1] = w;
#pragma unroll
    for (int kb = 2; kb <= R3 / 2; ++kb) pb[kb] = cmul(pb[kb - 1], w);
    {
      const cf wh = pb[R3 / 2];                  // w^(R3/2) -> index R3/2 holds w^(-R3/2)
#pragma unroll
      for (int kb = R3 / 2 + 1; kb < R3; ++kb) pb[kb] = make_float2(pb[R3 - kb].x, -pb[R3 - kb].y);
      pb[R3 / 2] = make_float2(wh.x, -wh.y);
    }
#pragma unroll
    for (int ul = 0; ul < P::U3L; ++ul) {
      const int wq = lane + 64 * (u0 + ul), k1 = wq / R2, ka = wq % R2;
      Dft<R3, false>::run(&v[ul * R3]);
#pragma unroll
      for (int kb = 0; kb < R3; ++kb) v[ul * R3 + kb] = cmul(v[ul * R3 + kb], cmul(pa, pb[kb]));
      // ---------------- inverse (DIT) ----------------
      Dft<R3, true>::run(&v[ul * R3]);
#pragma unroll
      for (int b = 0; b < R3; ++b) lds[k1 * P::T1 + ka * P::T2 + b] = v[ul * R3 + b];
      pa = cmul(pa, z);
    }
  }
  xbar<P>();
#pragma unroll
  for (int ul = 0; ul < P::U2L; ++ul) {
    const int t = lane + 64 * (sub * P::U2L + ul), k1 = t / M2, b = t % M2;
#pragma unroll
    for (int ka = 0; ka < R2; ++ka) v[ul * R2 + ka] = lds[k1 * P::T1 + ka * P::T2 + b];
  }
  xbar<P>();
#pragma unroll
  for (int ul = 0; ul < P::U2L; ++ul) {
    const int t = lane + 64 * (sub * P::U2L + ul), k1 = t / M2, b = t % M2;
#pragma unroll
    for (int ka = 1; ka < R2; ++ka) v[ul * R2 + ka] = tw.template apply2<true>(v[ul * R2 + ka], ka);
    Dft<R2, true>::run(&v[ul * R2]);
#pragma unroll
    for (int a = 0; a < R2; ++a) lds[k1 * P::S1 + M2 * a + b] = v[ul * R2 + a];
  }
  xbar<P>();
#pragma unroll
  for (int ul = 0; ul < P::U1L; ++ul) {
    const int u = sub * P::U1L + ul;
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) v[ul * R1 + k1] = lds[k1 * P::S1 + lane + 64 * u];
  }
  xbar<P>();
#pragma unroll
  for (int ul = 0; ul < P::U1L; ++ul) {
    const int u = sub * P::U1L + ul;
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1) v[ul * R1 + k1] = tw.template apply1<true>(v[ul * R1 + k1], u, k1);
    Dft<R1, true>::run(&v[ul * R1]);
  }
}

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
        v[ul * P::R1 + n1] = make_float2(val, 0.f);
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
          cf val = make_float2(0.f, 0.f);
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
    int64_t budget = ctx->opt("rot_ws_mb", 2048) * (int64_t)(1 << 20);
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
