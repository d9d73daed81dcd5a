// fft_wave.h -- wave-resident power-of-two FFT building blocks for the derotation kernels.
// A line of L = R1*R2*R3 complex values is distributed over WPL waves, L/(64*WPL) values per lane.
//   fft_forward : natural order (distribution D1) -> spectrum, digit-reversed (distribution D3), stage-3
//                 butterflies included:  v[ul*R3 + kb] = X[k1 + R1*ka + R1*R2*kb], (k1,ka) = divmod(lane+64u, R2)
//   fft_inverse : spectrum in D3 (already multiplied, 1/L folded in) -> natural order (D1)
// Four exchanges per forward+inverse pair through a padded LDS region shared by the WPL waves of the line.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <math.h>

namespace vipmi {
namespace fftw {

// complex = two packed floats in an even-aligned VGPR pair
typedef float cf __attribute__((ext_vector_type(2)));
__host__ __device__ __forceinline__ cf mkcf(float x, float y) { return cf{x, y}; }

// Complex arithmetic on the packed-FP32 pipe with VOP3P operand swizzles (op_sel / op_sel_hi pick the half of
// each 64-bit source feeding the low / high result lane, neg_lo / neg_hi negate it), so a complex multiply is
// exactly TWO instructions and multiplications by -i / +i fold into the following add -- hipcc's own lowering of
// the scalar formulas spends ~40 % of the FFT kernels on v_mov / v_xor / unpacked multiplies.
//   a*b       : t = (a.x b.x, a.x b.y) ; d = (a.y (-b.y) + t.x, a.y b.x + t.y)
//   a*conj(b) : t = (a.x b.x, -a.x b.y); d = (a.y b.y + t.x,   a.y b.x + t.y)
// (Measured and dropped, round 5: the two instructions as SEPARATE asm statements, so that the scheduler may put independent work
//  between the multiply and its dependent multiply-add -- the compiler pads every asm boundary instead (+250 .. +475 s_nop per line
//  pair) and the shears get slower: 1024 px 4.80 -> 5.20 ms per 100 frames, 512 px 3.87 -> 3.89 per 400.)
__device__ __forceinline__ cf cmul(cf a, cf b) {
  cf d;      // one asm statement: no compiler-inserted boundary pad between the two instructions
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "=&v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ cf cmulc(cf a, cf b) {
  cf d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
      : "=&v"(d) : "v"(a), "v"(b));
  return d;
}
// a*b + c in two instructions
__device__ __forceinline__ cf cmla(cf a, cf b, cf c) {
  cf d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// a + conj(b) and a - conj(b)
__device__ __forceinline__ cf cadd_conj(cf a, cf b) {
  cf d;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ cf csub_conj(cf a, cf b) {
  cf d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
// a + (-i) b = (a.x + b.y, a.y - b.x)   and   a - (-i) b = a + i b = (a.x - b.y, a.y + b.x)
// The swapped operand b sits in SRC0 (op_sel:[1,0]): with b in src1 (op_sel:[0,1]) gfx950 returns wrong lanes beside another
// wave's 16x16x64-i8 / 16x16x32-bf16 / f16 MFMA (common.h, VIPMI_NO_PK32); a + b = b + a bit for bit.
__device__ __forceinline__ cf add_mi(cf a, cf b) {
  cf d;
  asm("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ cf add_pi(cf a, cf b) {
  cf d;
  asm("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// multiply by -i (forward) or +i (inverse)
template <bool INV>
__device__ __forceinline__ cf mul_mi(cf a) { return INV ? mkcf(-a.y, a.x) : mkcf(a.y, -a.x); }
// a + (-i) b (forward) / a + i b (inverse), and the matching differences
template <bool INV>
__device__ __forceinline__ cf add_rot(cf a, cf b) { return INV ? add_pi(a, b) : add_mi(a, b); }
template <bool INV>
__device__ __forceinline__ cf sub_rot(cf a, cf b) { return INV ? add_mi(a, b) : add_pi(a, b); }

// w16^e = exp(-2 pi i e/16), e = 0..7
__device__ __forceinline__ cf w16(int e) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R = 0.70710678118654752f;
  switch (e & 7) {
    case 0: return mkcf(1.f, 0.f);
    case 1: return mkcf(C1, -S1);
    case 2: return mkcf(R, -R);
    case 3: return mkcf(S1, -C1);
    case 4: return mkcf(0.f, -1.f);
    case 5: return mkcf(-S1, -C1);
    case 6: return mkcf(-R, -R);
    default: return mkcf(-C1, -S1);
  }
}

template <bool INV>
__device__ __forceinline__ cf twc(cf v, int e16) {   // v * w16^e (forward) or v * conj(w16^e) (inverse)
  const cf w = w16(e16);
  return INV ? cmulc(v, w) : cmul(v, w);
}

template <bool INV>
__device__ __forceinline__ void dft4(cf& a0, cf& a1, cf& a2, cf& a3) {
  const cf t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), d = csub(a1, a3);
  a0 = cadd(t0, t2);
  a2 = csub(t0, t2);
  a1 = add_rot<INV>(t1, d);      // t1 + (-+i) d
  a3 = sub_rot<INV>(t1, d);      // t1 - (-+i) d
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
    o3 = twc<INV>(o3, 6);
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = add_rot<INV>(e2, o2); v[6] = sub_rot<INV>(e2, o2);     // w8^2 = -+i folded into the add
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
        else if (e == 9) { cf t = twc<INV>(y[n2][k1], 1); y[n2][k1] = mkcf(-t.x, -t.y); }  // w16^9 = -w16^1
        else y[n2][k1] = twc<INV>(y[n2][k1], e);
      }
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
      dft4<INV>(y[0][k1], y[1][k1], y[2][k1], y[3][k1]);
      v[k1] = y[0][k1]; v[k1 + 4] = y[1][k1]; v[k1 + 8] = y[2][k1]; v[k1 + 12] = y[3][k1];
    }
  }
};

// v * w16^E (forward) or v * conj(w16^E) (inverse), E a compile-time exponent
template <bool INV, int E>
__device__ __forceinline__ cf mulw16(cf v) {
  constexpr int e = E & 15;
  if constexpr (e == 0) return v;
  else if constexpr (e == 8) return mkcf(-v.x, -v.y);
  else if constexpr (e == 4) return mul_mi<INV>(v);              // w16^4 = -i
  else if constexpr (e == 12) return mul_mi<!INV>(v);            // w16^12 = +i
  else if constexpr (e < 8) return twc<INV>(v, e);
  else {
    const cf t = twc<INV>(v, e - 8);
    return mkcf(-t.x, -t.y);
  }
}

// Pruned radix-16 butterflies.  A frame of N = L/4 pixels sits at canvas offset 3L/8, so of the 16 inputs of a first-stage
// (decimation-in-frequency) butterfly only n1 = 6..9 are non-zero, and of the 16 outputs of a last-stage inverse
// butterfly only n1 = 6..9 survive the crop.  With k = l + 4h:
//   X[l + 4h] = sum_m a_m w16^((6+m)(l+4h)) = (-1)^h sum_m [a_m w16^((6+m) l)] w4^(hm)
// i.e. four 4-point DFTs of twiddled inputs; the factor (-1)^h = w4^(2h) is a circular shift of the DFT4 input by two.
// 4 DFT4 + 8 non-trivial twiddles instead of 8 DFT4 + 8 twiddles (forward), + 12 adds (inverse).
__device__ __forceinline__ void dft16_fwd_live4(cf* __restrict__ v) {      // in: v[6..9]; out: v[0..15] natural order
  const cf a0 = v[6], a1 = v[7], a2 = v[8], a3 = v[9];
#define VIPMI_P16_FWD(l)                                                                             \
  {                                                                                                  \
    cf y0 = mulw16<false, 6 * l>(a0), y1 = mulw16<false, 7 * l>(a1), y2 = mulw16<false, 8 * l>(a2),  \
       y3 = mulw16<false, 9 * l>(a3);                                                                \
    dft4<false>(y2, y3, y0, y1);                                                                     \
    v[l] = y2; v[l + 4] = y3; v[l + 8] = y0; v[l + 12] = y1;                                         \
  }
  VIPMI_P16_FWD(0) VIPMI_P16_FWD(1) VIPMI_P16_FWD(2) VIPMI_P16_FWD(3)
#undef VIPMI_P16_FWD
}
__device__ __forceinline__ void dft16_inv_keep4(cf* __restrict__ v) {      // in: v[0..15]; out: v[6..9] only
  cf x0, x1, x2, x3;
#define VIPMI_P16_INV(l)                                                                             \
  {                                                                                                  \
    cf z0 = v[l], z1 = v[l + 4], z2 = v[l + 8], z3 = v[l + 12];                                      \
    dft4<true>(z0, z1, z2, z3);                                                                      \
    const cf t0 = mulw16<true, 6 * l>(z2), t1 = mulw16<true, 7 * l>(z3), t2 = mulw16<true, 8 * l>(z0), \
             t3 = mulw16<true, 9 * l>(z1);                                                           \
    if (l == 0) { x0 = t0; x1 = t1; x2 = t2; x3 = t3; }                                              \
    else { x0 = cadd(x0, t0); x1 = cadd(x1, t1); x2 = cadd(x2, t2); x3 = cadd(x3, t3); }             \
  }
  VIPMI_P16_INV(0) VIPMI_P16_INV(1) VIPMI_P16_INV(2) VIPMI_P16_INV(3)
#undef VIPMI_P16_INV
  v[6] = x0; v[7] = x1; v[8] = x2; v[9] = x3;
}

template <int R1_, int R2_, int R3_, int S1_, int T1_, int T2_, int WPB_, int WPL_, bool TW2R_ = false>
struct Plan {
  static constexpr bool TW2R = TW2R_;                   // composed stage-2 twiddles live in registers (R2 - 1 complex)
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
  static constexpr bool CAN_PRUNE = (R1_ == 16);         // pruned first / last radix-16 stage (dft16_fwd_live4 / dft16_inv_keep4)
  static_assert(U1 * R1 == VPT && U2 * R2 == VPT && U3 * R3 == VPT, "bad radix plan");
};
// WPL > 1: the line is split over WPL waves (VL = 16 complex registers per lane instead of 32/64), which
// keeps the kernels under 128 VGPRs (4 waves/SIMD, no spills); the exchanges then need a workgroup barrier.
using Plan512 = Plan<8, 8, 8, 72, 72, 9, 8, 1>;
using Plan1024 = Plan<16, 8, 8, 72, 72, 9, 8, 1, true>;
using Plan1024q = Plan<16, 8, 8, 72, 72, 9, 4, 1, true>;   // four lines per workgroup: three workgroups per CU = THREE waves per SIMD (the kernels hold <= 164 VGPRs)
using Plan2048w1 = Plan<16, 16, 8, 136, 152, 9, 8, 1, true>;   // one wave per line: no workgroup barriers, 2 waves/SIMD
using Plan2048w1h = Plan<16, 16, 8, 136, 152, 9, 4, 1, true>;  // the same with 4 lines per workgroup (ONE wave per SIMD): the per-frame K kernel rs_aux_k runs on it
using Plan4096w2 = Plan<16, 16, 16, 272, 272, 17, 8, 2>;    // two waves per line, 2 waves/SIMD, 256-VGPR budget
using Plan4096w1 = Plan<16, 16, 16, 272, 272, 17, 4, 1, true>;   // one wave per line, ONE wave per SIMD (512-VGPR budget)

// exp(-2 pi i j/64), j = 0..63
__device__ const float ROOT64_C[64] = {1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f, 6.123233996e-17f, -9.801714033e-02f, -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f, -4.713967368e-01f, -5.555702330e-01f, -6.343932842e-01f, -7.071067812e-01f, -7.730104534e-01f, -8.314696123e-01f, -8.819212643e-01f, -9.238795325e-01f, -9.569403357e-01f, -9.807852804e-01f, -9.951847267e-01f, -1.000000000e+00f, -9.951847267e-01f, -9.807852804e-01f, -9.569403357e-01f, -9.238795325e-01f, -8.819212643e-01f, -8.314696123e-01f, -7.730104534e-01f, -7.071067812e-01f, -6.343932842e-01f, -5.555702330e-01f, -4.713967368e-01f, -3.826834324e-01f, -2.902846773e-01f, -1.950903220e-01f, -9.801714033e-02f, -1.836970199e-16f, 9.801714033e-02f, 1.950903220e-01f, 2.902846773e-01f, 3.826834324e-01f, 4.713967368e-01f, 5.555702330e-01f, 6.343932842e-01f, 7.071067812e-01f, 7.730104534e-01f, 8.314696123e-01f, 8.819212643e-01f, 9.238795325e-01f, 9.569403357e-01f, 9.807852804e-01f, 9.951847267e-01f};
__device__ const float ROOT64_S[64] = {-0.000000000e+00f, -9.801714033e-02f, -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f, -4.713967368e-01f, -5.555702330e-01f, -6.343932842e-01f, -7.071067812e-01f, -7.730104534e-01f, -8.314696123e-01f, -8.819212643e-01f, -9.238795325e-01f, -9.569403357e-01f, -9.807852804e-01f, -9.951847267e-01f, -1.000000000e+00f, -9.951847267e-01f, -9.807852804e-01f, -9.569403357e-01f, -9.238795325e-01f, -8.819212643e-01f, -8.314696123e-01f, -7.730104534e-01f, -7.071067812e-01f, -6.343932842e-01f, -5.555702330e-01f, -4.713967368e-01f, -3.826834324e-01f, -2.902846773e-01f, -1.950903220e-01f, -9.801714033e-02f, -1.224646799e-16f, 9.801714033e-02f, 1.950903220e-01f, 2.902846773e-01f, 3.826834324e-01f, 4.713967368e-01f, 5.555702330e-01f, 6.343932842e-01f, 7.071067812e-01f, 7.730104534e-01f, 8.314696123e-01f, 8.819212643e-01f, 9.238795325e-01f, 9.569403357e-01f, 9.807852804e-01f, 9.951847267e-01f, 1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f};
__device__ __forceinline__ cf root64(int j) { return mkcf(ROOT64_C[j & 63], ROOT64_S[j & 63]); }

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
  static constexpr int LDS_ELEMS = PER_LANE * 64 + 8;      // table + 16 barrier counters (64 B)
  const cf* tab;                                          // LDS, already offset by lane
  cf w2[P::TW2R ? P::R2 - 1 : 1];                         // TW2R: w_M1^(b ka), ka = 1 .. R2-1, composed once per kernel
  unsigned* bar;                                          // LDS arrival counter of this line slot (WPL > 1)
  mutable unsigned bar_target;
  static constexpr int BAR_ELEMS = 8;                     // LDS reserved for the counters, in cf units (64 B)

  __device__ __forceinline__ void init(const cf* __restrict__ gtab, cf* __restrict__ ltab, int lane) {
    for (int e = threadIdx.x; e < PER_LANE * 64; e += blockDim.x) {
      const int ln = e / PER_LANE, j = e % PER_LANE;
      ltab[j * 64 + ln] = gtab[e];
    }
    unsigned* counters = reinterpret_cast<unsigned*>(ltab + PER_LANE * 64);
    if (threadIdx.x < 16) counters[threadIdx.x] = 0;
    __syncthreads();
    tab = ltab + lane;
    bar = counters + (threadIdx.x >> 6) / P::WPL;
    bar_target = 0;
    if constexpr (P::TW2R) {
#pragma unroll
      for (int ka = 1; ka < P::R2; ++ka) {
        const int h = ka >> 2, l = ka & 3;
        w2[ka - 1] = (h && l) ? cmul(t2h(h), t2l(l)) : (h ? t2h(h) : t2l(l));
      }
    }
  }
  static void fill_table(std::vector<cf>& tabv) {
    tabv.resize(64 * PER_LANE);
    auto unit = [](long e, long period) {
      const double ang = -2.0 * M_PI * (double)(e % period) / (double)period;
      return mkcf((float)cos(ang), (float)sin(ang));
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
    if constexpr (P::TW2R) return CONJ ? cmulc(v, w2[ka - 1]) : cmul(v, w2[ka - 1]);
    const int h = ka >> 2, l = ka & 3;
    cf w;
    if (h && l) w = cmul(t2h(h), t2l(l));
    else if (h) w = t2h(h);
    else w = t2l(l);
    return CONJ ? cmulc(v, w) : cmul(v, w);
  }
};

// Barrier between the exchange phases of one line.
//   WPL == 1 : the line is wave-private -> LDS ops of a wave are in order, only the compiler must not reorder
//   WPL  > 1 : the WPL waves of the line synchronise through a counter in LDS (arrive = ds_add, wait = poll).
//              A workgroup-wide s_barrier would work too but locks all lines of the workgroup into the same
//              phase (every wave in its LDS phase, then every wave in its VALU phase); the pair-level barrier
//              lets the lines of a workgroup drift apart so LDS, VALU and memory phases overlap.
template <class P>
__device__ __forceinline__ void xbar(const Twiddles<P>& tw) {
  if (P::WPL == 1) {
    __builtin_amdgcn_wave_barrier();
    return;
  }
  // (an LDS-counter barrier private to the WPL waves of a line was tried: the polling loop costs more than
  //  the lockstep it removes -- 1.62 vs 1.41 ms per 100 frames for the column shear -- so use s_barrier)
  __syncthreads();
}

// forward transform: D1 in, D3 (spectrum) out
// LIVE4: only the inputs n1 = 6..9 of every first-stage butterfly are set (and non-zero); the others are not read
template <class P, bool LIVE4 = false>
__device__ __forceinline__ void fft_forward(cf (&v)[P::VL], const Twiddles<P>& tw, cf* __restrict__ lds, int lane,
                                            int sub) {
  static_assert(!LIVE4 || P::CAN_PRUNE, "pruned first stage needs R1 = 16");
  constexpr int R1 = P::R1, R2 = P::R2, R3 = P::R3, M2 = P::M2;
  // ---------------- forward (DIF) ----------------
#pragma unroll
  for (int ul = 0; ul < P::U1L; ++ul) {
    const int u = sub * P::U1L + ul;
    if constexpr (LIVE4) dft16_fwd_live4(&v[ul * R1]);
    else Dft<R1, false>::run(&v[ul * R1]);
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1) v[ul * R1 + k1] = tw.template apply1<false>(v[ul * R1 + k1], u, k1);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) lds[k1 * P::S1 + lane + 64 * u] = v[ul * R1 + k1];
  }
  xbar<P>(tw);
#pragma unroll
  for (int ul = 0; ul < P::U2L; ++ul) {
    const int t = lane + 64 * (sub * P::U2L + ul), k1 = t / M2, b = t % M2;
#pragma unroll
    for (int a = 0; a < R2; ++a) v[ul * R2 + a] = lds[k1 * P::S1 + M2 * a + b];
  }
  xbar<P>(tw);
#pragma unroll
  for (int ul = 0; ul < P::U2L; ++ul) {
    const int t = lane + 64 * (sub * P::U2L + ul), k1 = t / M2, b = t % M2;
    Dft<R2, false>::run(&v[ul * R2]);
#pragma unroll
    for (int ka = 1; ka < R2; ++ka) v[ul * R2 + ka] = tw.template apply2<false>(v[ul * R2 + ka], ka);
#pragma unroll
    for (int ka = 0; ka < R2; ++ka) lds[k1 * P::T1 + ka * P::T2 + b] = v[ul * R2 + ka];
  }
  xbar<P>(tw);
#pragma unroll
  for (int ul = 0; ul < P::U3L; ++ul) {
    const int w = lane + 64 * (sub * P::U3L + ul), k1 = w / R2, ka = w % R2;
#pragma unroll
    for (int b = 0; b < R3; ++b) v[ul * R3 + b] = lds[k1 * P::T1 + ka * P::T2 + b];
  }
  xbar<P>(tw);
#pragma unroll
  for (int ul = 0; ul < P::U3L; ++ul) Dft<R3, false>::run(&v[ul * R3]);
}

// inverse transform: spectrum in D3 (v[ul*R3 + kb]) -> D1 natural order; unnormalised (fold 1/L into the multiplier)
// KEEP4: only the outputs n1 = 6..9 of every last-stage butterfly are produced (the crop window)
template <class P, bool KEEP4 = false>
__device__ __forceinline__ void fft_inverse(cf (&v)[P::VL], const Twiddles<P>& tw, cf* __restrict__ lds, int lane,
                                            int sub) {
  static_assert(!KEEP4 || P::CAN_PRUNE, "pruned last stage needs R1 = 16");
  constexpr int R1 = P::R1, R2 = P::R2, R3 = P::R3, M2 = P::M2;
#pragma unroll
  for (int ul = 0; ul < P::U3L; ++ul) {
    const int wq = lane + 64 * (sub * P::U3L + ul), k1 = wq / R2, ka = wq % R2;
    Dft<R3, true>::run(&v[ul * R3]);
#pragma unroll
    for (int b = 0; b < R3; ++b) lds[k1 * P::T1 + ka * P::T2 + b] = v[ul * R3 + b];
  }
  xbar<P>(tw);
#pragma unroll
  for (int ul = 0; ul < P::U2L; ++ul) {
    const int t = lane + 64 * (sub * P::U2L + ul), k1 = t / M2, b = t % M2;
#pragma unroll
    for (int ka = 0; ka < R2; ++ka) v[ul * R2 + ka] = lds[k1 * P::T1 + ka * P::T2 + b];
  }
  xbar<P>(tw);
#pragma unroll
  for (int ul = 0; ul < P::U2L; ++ul) {
    const int t = lane + 64 * (sub * P::U2L + ul), k1 = t / M2, b = t % M2;
#pragma unroll
    for (int ka = 1; ka < R2; ++ka) v[ul * R2 + ka] = tw.template apply2<true>(v[ul * R2 + ka], ka);
    Dft<R2, true>::run(&v[ul * R2]);
#pragma unroll
    for (int a = 0; a < R2; ++a) lds[k1 * P::S1 + M2 * a + b] = v[ul * R2 + a];
  }
  xbar<P>(tw);
#pragma unroll
  for (int ul = 0; ul < P::U1L; ++ul) {
    const int u = sub * P::U1L + ul;
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) v[ul * R1 + k1] = lds[k1 * P::S1 + lane + 64 * u];
  }
  xbar<P>(tw);
#pragma unroll
  for (int ul = 0; ul < P::U1L; ++ul) {
    const int u = sub * P::U1L + ul;
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1) v[ul * R1 + k1] = tw.template apply1<true>(v[ul * R1 + k1], u, k1);
    if constexpr (KEEP4) dft16_inv_keep4(&v[ul * R1]);
    else Dft<R1, true>::run(&v[ul * R1]);
  }
}

// exp(-2 pi i * turns), argument reduced in float64
__device__ __forceinline__ cf expi_turns(double turns) {
  turns -= rint(turns);
  float sn, cs;
  sincospif((float)(-2.0 * turns), &sn, &cs);
  return mkcf(cs, sn);
}

// Shear phase of one line: exp(-2 pi i k s/L)/L for k = k1 + R1*ka + R1*R2*kb (signed through kb) factorises as
// pa(u) * pb(kb):  pa(u0) per lane, pa(u+1) = pa(u)*z,  pb(kb) = w^kbs  (powers 0..R3/2 kept; negatives = conj).
template <class P>
struct ShearPhase {
  cf pa, z;
  cf wp[P::R3 / 2 + 1];
  __device__ __forceinline__ void init(double s, int lane, int sub) {
    const double sl = s / (double)P::L;
    const cf w = expi_turns((double)(P::R1 * P::R2) * sl);
    z = expi_turns((double)(64 / P::R2) * sl);
    const int u0 = sub * P::U3L;
    pa = expi_turns((double)(lane / P::R2 + (64 / P::R2) * u0 + P::R1 * (lane % P::R2)) * sl);
    pa = mkcf(pa.x * (1.0f / (float)P::L), pa.y * (1.0f / (float)P::L));
    wp[0] = mkcf(1.f, 0.f);
#pragma unroll
    for (int i = 1; i <= P::R3 / 2; ++i) wp[i] = cmul(wp[i - 1], w);
  }
  // phases of the shift s + delta from those of s: d_lane = exp(-2 pi i m_lane delta/L) etc. (pa keeps its scale)
  __device__ __forceinline__ void init_from(const ShearPhase& o, cf d_lane, cf d_z, cf d_w) {
    pa = cmul(o.pa, d_lane);
    z = cmul(o.z, d_z);
    const cf w = cmul(o.wp[1], d_w);
    wp[0] = mkcf(1.f, 0.f);
#pragma unroll
    for (int i = 1; i <= P::R3 / 2; ++i) wp[i] = cmul(wp[i - 1], w);
  }
  __device__ __forceinline__ cf pb(int kb) const {          // kb compile-time after unrolling
    return kb < P::R3 / 2 ? wp[kb] : mkcf(wp[P::R3 - kb].x, -wp[P::R3 - kb].y);
  }
  __device__ __forceinline__ void next_u() { pa = cmul(pa, z); }
};

// y = ifft(fft(x) * exp(-2 pi i f s)): complex line, D1 in / D1 out
template <class P>
__device__ __forceinline__ void line_shift(cf (&v)[P::VL], const Twiddles<P>& tw, cf* __restrict__ lds,
                                           double s, int lane, int sub) {
  fft_forward<P>(v, tw, lds, lane, sub);
  ShearPhase<P> ph;
  ph.init(s, lane, sub);
#pragma unroll
  for (int ul = 0; ul < P::U3L; ++ul) {
#pragma unroll
    for (int kb = 0; kb < P::R3; ++kb) v[ul * P::R3 + kb] = cmul(v[ul * P::R3 + kb], cmul(ph.pa, ph.pb(kb)));
    ph.next_u();
  }
  fft_inverse<P>(v, tw, lds, lane, sub);
}

}  // namespace fftw
}  // namespace vipmi
