// collapse.hip -- cube_collapse (preproc/subsampling.py:30-116): per-pixel NaN-aware reduction
// over the n frames of a cube[n,P].
//
//  * mean / sum / max / absmean / wmean: one thread per pixel column, coalesced streaming read of
//    the cube (HBM-bound: n*P*4 bytes in, P*4 out).
//  * median (nanmedian): a workgroup stages a [n frames][TP pixels] tile in LDS with coalesced
//    128-byte row segments (row stride TP+1 floats: the transposed read is bank-conflict free);
//    then ONE WAVE PER PIXEL holds the pixel's n values in registers (n/64 per lane) and selects
//    the middle one(s) by bucket selection (median_vals below: 256 linear bins between the pixel's
//    minimum and maximum, exact finish on the <= 64 samples of the bin the rank falls into); the
//    32-step bitwise bisection on order-preserving uint32 keys of round 1 (RPL v_cmp + wave ballots
//    per step) is the fallback and serves trimmean.  For an even number of valid samples the upper
//    median is the smallest sample above the lower one (or the lower one again if it is duplicated)
//    and the result is (a+b)*0.5 in float32, as numpy computes it.
#include "common.h"

namespace vipmi {

namespace {

__device__ __forceinline__ unsigned f2key(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// Wave-wide minimum on DPP (xor butterflies inside a 16-lane row, then the four row minima through readlane); the
// ds_bpermute-based __shfl_xor costs an LDS round trip per step.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned umin_(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  v = umin_(v, dpp_u32<0xB1>(v));     // quad_perm [1,0,3,2]
  v = umin_(v, dpp_u32<0x4E>(v));     // quad_perm [2,3,0,1]
  v = umin_(v, dpp_u32<0x141>(v));    // row_half_mirror
  v = umin_(v, dpp_u32<0x140>(v));    // row_mirror
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return umin_(umin_(a, b), umin_(c, d));
}

// Wave-wide sum of a small per-lane count (0 .. RPL): one ballot per bit of the count.
template <int RPL>
__device__ __forceinline__ int wave_count(int c) {
  int tot = __popcll(__ballot(c & 1));
  if (RPL >= 2) tot += 2 * __popcll(__ballot(c & 2));
  if (RPL >= 4) tot += 4 * __popcll(__ballot(c & 4));
  if (RPL >= 8) tot += 8 * __popcll(__ballot(c & 8));
  if (RPL >= 16) tot += 16 * __popcll(__ballot(c & 16));
  if (RPL >= 32) tot += 32 * __popcll(__ballot(c & 32));
  if (RPL >= 64) tot += 64 * __popcll(__ballot(c & 64));
  return tot;
}

// rank-k (0-based) order statistic of the wave's keys: 32-step bitwise bisection.  The RPL comparisons of a step
// are counted per lane on the vector ALU and summed across the wave with log2(RPL)+1 ballots (a ballot per key
// serialises RPL VALU->SALU round trips per step, which made the median kernel latency bound).
template <int RPL>
__device__ __forceinline__ unsigned select_rank(const unsigned (&key)[RPL], int k) {
  unsigned ans = 0;
  for (int b = 31; b >= 0; --b) {
    const unsigned cand = ans | (1u << b);
    int c = 0;
#pragma unroll
    for (int r = 0; r < RPL; ++r) c += (key[r] < cand) ? 1 : 0;
    if (wave_count<RPL>(c) <= k) ans = cand;
  }
  return ans;
}

// ---- median by bucket selection -------------------------------------------------------------------------------------
// The 32-step bisection above costs ~830 VALU instructions per pixel (98 % VALU busy, rocprofv3 SQ counters of round 2).
// For the median itself: ONE pass that bins the wave's keys linearly in VALUE between their minimum and maximum (256
// bins, LDS histogram private to the wave; the binning is a weakly monotone function of the key, so the bin in which the
// cumulative count crosses the wanted rank holds the wanted element), a DPP prefix sum over the bins, and an exact finish
// on the <= 64 keys of that bin: one key per lane, rank by counting against every candidate (readlane broadcasts).  A
// bin with more than 64 keys (outliers stretching the range, ties) is binned again over its own range, at most 3
// levels; then the bisection takes over.  Every comparison that decides a rank is made on the order-preserving keys.
__device__ __forceinline__ unsigned umax_(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = umax_(v, dpp_u32<0xB1>(v));
  v = umax_(v, dpp_u32<0x4E>(v));
  v = umax_(v, dpp_u32<0x141>(v));
  v = umax_(v, dpp_u32<0x140>(v));
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return umax_(umax_(a, b), umax_(c, d));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_rows(unsigned v) {      // lanes of rows outside ROWMASK (and invalid sources) read 0
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false);
}
__device__ __forceinline__ unsigned wave_inclusive_sum(unsigned v) {
  v += dpp_rows<0x111, 0xf>(v);      // row_shr:1
  v += dpp_rows<0x112, 0xf>(v);      // row_shr:2
  v += dpp_rows<0x114, 0xf>(v);      // row_shr:4
  v += dpp_rows<0x118, 0xf>(v);      // row_shr:8   -> inclusive sums inside every row of 16 lanes
  v += dpp_rows<0x142, 0xa>(v);      // row_bcast:15 -> rows 1 and 3 add the total of the row before
  v += dpp_rows<0x143, 0xc>(v);      // row_bcast:31 -> rows 2 and 3 add the total of rows 0..1
  return v;
}
// Lanes of one wave exchanging data through LDS: the hardware executes a wave's LDS instructions in order, but the
// COMPILER needs the fences -- without them it forwards a lane's own earlier store to its later load across the other
// lanes' atomics / stores (it sank the histogram read into the divergent branch of the lanes that had added to it).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int lane_rank_in(unsigned long long mask) {   // set bits of `mask` below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// ---- the selection in the FLOAT domain (round 5) ---------------------------------------------------------------------------
// Round 4's selection carried order-preserving uint32 keys: 3 operations per sample to form the key, 3 to turn it back into the
// value the bins are linear in, 3 to keep the 0xffffffff of a NaN out of the maximum, two unsigned range tests per sample and level
// -- 25 vector instructions per sample row where the arithmetic needs 15 (the kernel is issue-bound: 384 per pixel at n = 400).
// Here a sample stays the float it is.  NaNs (quieted when the sample is read) drop out of v_min_f32 / v_max_f32 by themselves (IEEE
// minNum / maxNum: the other operand), fail every ordered comparison -- so they never enter a bin range, a count or a candidate
// list -- and the m valid samples are the ones an order statistic of rank < m is taken from, as with the keys.  Equal floats are
// equal keys except for -0.0 / +0.0, which compare equal here: the result can differ from the key version in the sign of a zero
// only (numpy's partition does not order them either).
__device__ __forceinline__ float fmin_raw(float a, float b) {      // v_min_f32 itself (fminf adds a canonicalising v_max per operand)
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float fmax_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// (kernels that carry VIPMI_NO_PK32 cannot inline the HIP header's plain-inline functions -- different target features -- and
// would CALL __ballot / __popcll / __uint_as_float; inside a forceinline helper they are inlined bottom-up first)
__device__ __forceinline__ void block_sync() { __syncthreads(); }
__device__ __forceinline__ int lanes_true(bool p) { return (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(p)); }
__device__ __forceinline__ float quiet_nan() { return __uint_as_float(0x7fc00000u); }
// Wave reductions of floats that are never NaN: v_min_f32 / v_max_f32 with the DPP operand directly (the compiler can fold a DPP move
// only into an operation it generates itself, and its fminf carries a canonicalising v_max per operand; through the key domain a
// reduction cost 20 vector instructions, four per pixel).  Six steps: inside quads, inside rows of 16, then row_bcast:15 / :31 carry
// the row results up to lane 63.  A DPP source written by the previous vector instruction needs two wait states: the two
// reductions of a pair are interleaved, one s_nop between steps.
#define VIPMI_DPP_ALL " row_mask:0xf bank_mask:0xf\n\t"
#define VIPMI_RED2(OPA, OPB)                                                                                                  \
  "s_nop 1\n\t"                                                                                                              \
  OPA " %0, %0, %0 quad_perm:[1,0,3,2]" VIPMI_DPP_ALL OPB " %1, %1, %1 quad_perm:[1,0,3,2]" VIPMI_DPP_ALL "s_nop 0\n\t"       \
  OPA " %0, %0, %0 quad_perm:[2,3,0,1]" VIPMI_DPP_ALL OPB " %1, %1, %1 quad_perm:[2,3,0,1]" VIPMI_DPP_ALL "s_nop 0\n\t"       \
  OPA " %0, %0, %0 row_half_mirror" VIPMI_DPP_ALL OPB " %1, %1, %1 row_half_mirror" VIPMI_DPP_ALL "s_nop 0\n\t"               \
  OPA " %0, %0, %0 row_mirror" VIPMI_DPP_ALL OPB " %1, %1, %1 row_mirror" VIPMI_DPP_ALL "s_nop 0\n\t"                         \
  OPA " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" OPB " %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
  "s_nop 0\n\t"                                                                                                              \
  OPA " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" OPB " %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf"
#define VIPMI_RED1(OP)                                                                                                        \
  "s_nop 1\n\t"                                                                                                              \
  OP " %0, %0, %0 quad_perm:[1,0,3,2]" VIPMI_DPP_ALL "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1]" VIPMI_DPP_ALL         \
  "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror" VIPMI_DPP_ALL "s_nop 1\n\t" OP " %0, %0, %0 row_mirror" VIPMI_DPP_ALL        \
  "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                                                 \
  "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
__device__ __forceinline__ float lane63(float v) { return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63)); }
__device__ __forceinline__ float wave_min_f32(float v) {
  asm(VIPMI_RED1("v_min_f32_dpp") : "+v"(v));
  return lane63(v);
}
__device__ __forceinline__ float wave_max_f32(float v) {
  asm(VIPMI_RED1("v_max_f32_dpp") : "+v"(v));
  return lane63(v);
}
__device__ __forceinline__ void wave_min_max_f32(float& a, float& b) {        // a <- wave minimum of a, b <- wave maximum of b
  asm(VIPMI_RED2("v_min_f32_dpp", "v_max_f32_dpp") : "+v"(a), "+v"(b));
  a = lane63(a);
  b = lane63(b);
}
__device__ __forceinline__ void wave_max_max_f32(float& a, float& b) {
  asm(VIPMI_RED2("v_max_f32_dpp", "v_max_f32_dpp") : "+v"(a), "+v"(b));
  a = lane63(a);
  b = lane63(b);
}

// bins of one level: linear in value over [lo, hi]; everything outside (NaN included) to the lane's own dump word
template <int RPL, bool FIRST>
__device__ __forceinline__ void bin_level(const float (&v)[RPL], float lo, float hi, float scale, int lane, int (&bin)[RPL]) {
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    const int b = (int)((v[r] - lo) * scale);   // (in range: 0 .. 255)
    // first level: [lo, hi] holds every valid sample, so "in range" is "not NaN"
    const bool in = FIRST ? (v[r] == v[r]) : (v[r] >= lo && v[r] <= hi);
    bin[r] = in ? b : 257 + lane;
  }
}

// Order statistics of the valid samples of a pixel (the median: rank k = (m-1)/2 of the m valid samples and, for even m, the
// upper one, rank k+1; the trimmed mean: the two ends of its slice).  v: RPL samples per lane, NaN = no sample (quiet NaNs only).
// hist: HIST_WORDS words of LDS private to the wave: 256 bins, then (from word 257) one dump word per lane for the samples outside
// the current range (keeps the loops free of divergent branches), then 64 dump slots for the lanes that have no candidate to store
template <int RPL>
__device__ __forceinline__ void rank_vals(const float (&v)[RPL], int k, bool even, unsigned* __restrict__ hist, int lane, float& vlow,
                                          float& vhigh) {
  // vlow = the valid sample of rank k (0-based, k < number of valid samples); vhigh = the one of rank k + 1 when `even` (it exists)
  const float inf = __builtin_inff();
  float lo = inf, hi = -inf;
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    lo = fmin_raw(lo, v[r]);
    hi = fmax_raw(hi, v[r]);
  }
  wave_min_max_f32(lo, hi);
  int rank = k;                                  // rank of the wanted sample among the samples in [lo, hi]
  bool done = false;
  for (int level = 0; level < 3 && !done; ++level) {
    if (lo == hi) {                              // every remaining sample is the same value
      vlow = lo;
      vhigh = lo;
      done = true;
      if (even) {                                // upper median: the same value again, or the smallest sample above it
        int cle = 0;
        float nxt = inf;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          cle += (v[r] <= lo) ? 1 : 0;
          nxt = (v[r] > lo) ? fmin_raw(nxt, v[r]) : nxt;
        }
        if (wave_count<RPL>(cle) < k + 2) vhigh = wave_min_f32(nxt);
      }
      break;
    }
    // (an approximate reciprocal will do: the binning only has to be one weakly monotone function for every sample of the level)
    // 255.9 instead of 256: (hi - lo) * scale stays below 256 whatever the last bits of the reciprocal, so no bin needs clipping
    const float scale = 255.9f * __builtin_amdgcn_rcpf(hi - lo);
    if (!(scale > 0.f && scale < 3.0e38f)) break;          // range overflows / underflows: bisection
    reinterpret_cast<uint4*>(hist)[lane] = make_uint4(0u, 0u, 0u, 0u);
    wave_lds_sync();
    // bin of every sample (>= 257: outside the range) and, from the returning atomic, its ordinal inside the bin.
    // Outside the range: a dump word of the lane's own (257 + lane).  ONE shared dump bin made every atomic of a second level
    // a 64-way same-address collision (at level >= 1 nearly all samples are outside the range): real residual cubes, where a fifth
    // of the pixels need a second level, paid for it -- C5 10.5 ms against 5.8 on Gaussian noise (tools/time_median_c5.py)
    int bin[RPL];
    unsigned ord[RPL];
    if (level == 0) bin_level<RPL, true>(v, lo, hi, scale, lane, bin);
    else bin_level<RPL, false>(v, lo, hi, scale, lane, bin);
#pragma unroll
    for (int r = 0; r < RPL; ++r) ord[r] = atomicAdd(&hist[bin[r]], 1u);
    wave_lds_sync();
    const uint4 h = reinterpret_cast<const uint4*>(hist)[lane];
    const unsigned s4 = h.x + h.y + h.z + h.w;
    const unsigned incl = wave_inclusive_sum(s4);
    const unsigned long long above = __ballot(incl > (unsigned)rank);
    const int L = __builtin_ctzll(above);        // (never empty: the range holds more than `rank` samples)
    const unsigned hx = (unsigned)__builtin_amdgcn_readlane((int)h.x, L), hy = (unsigned)__builtin_amdgcn_readlane((int)h.y, L);
    const unsigned hz = (unsigned)__builtin_amdgcn_readlane((int)h.z, L), hw = (unsigned)__builtin_amdgcn_readlane((int)h.w, L);
    unsigned rem = (unsigned)rank - ((unsigned)__builtin_amdgcn_readlane((int)incl, L) - (hx + hy + hz + hw));
    int j = 0;
    unsigned c = hx;
    if (rem >= c) { rem -= c; j = 1; c = hy; }
    if (j == 1 && rem >= c) { rem -= c; j = 2; c = hz; }
    if (j == 2 && rem >= c) { rem -= c; j = 3; c = hw; }
    const int bstar = 4 * L + j;
    rank = (int)rem;
    wave_lds_sync();
    if (c <= 64u) {
      // the bin's samples go to slots 0..c-1 of the (dead) histogram by their ordinals, everything else to the dump slots
#pragma unroll
      for (int r = 0; r < RPL; ++r) hist[bin[r] == bstar ? ord[r] : 257u + (unsigned)lane] = __float_as_uint(v[r]);
      wave_lds_sync();
      const bool mine = (unsigned)lane < c;
      const float cand = __uint_as_float(hist[mine ? lane : 0]);
      int less = 0;
      for (unsigned q = 0; q < c; ++q) {
        const float vq = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(cand), (int)q));
        less += (vq < cand) ? 1 : 0;
      }
      // largest sample with at most `rank` samples below it -- and, while at it, with at most rank + 1 (the upper median when it
      // lies in this bin)
      float a = mine && less <= rank ? cand : -inf, b = mine && less <= rank + 1 ? cand : -inf;
      wave_max_max_f32(a, b);
      vlow = a;
      vhigh = vlow;
      if (even) {
        if ((unsigned)rank + 1u < c) {
          vhigh = b;
        } else {                                 // the next sample lives in a later bin: the smallest one above vlow
          float nxt = inf;
#pragma unroll
          for (int r = 0; r < RPL; ++r) nxt = (v[r] > vlow) ? fmin_raw(nxt, v[r]) : nxt;
          vhigh = wave_min_f32(nxt);
        }
      }
      wave_lds_sync();
      done = true;
    } else {
      // crowded bin: its own value range becomes the next level's range
      float nlo = inf, nhi = -inf;
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        if (bin[r] == bstar) {
          nlo = fmin_raw(nlo, v[r]);
          nhi = fmax_raw(nhi, v[r]);
        }
      }
      wave_min_max_f32(nlo, nhi);
      lo = nlo;
      hi = nhi;
    }
  }
  if (!done) {                                   // bisection on the order-preserving keys of all samples (global rank k)
    unsigned key[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) key[r] = (v[r] == v[r]) ? f2key(v[r]) : 0xffffffffu;
    const unsigned klow = select_rank<RPL>(key, k);
    unsigned khigh = klow;
    if (even) {
      int cle = 0;
      unsigned nxt = 0xffffffffu;
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        cle += (key[r] <= klow) ? 1 : 0;
        if (key[r] > klow && key[r] < nxt) nxt = key[r];
      }
      if (wave_count<RPL>(cle) < k + 2) khigh = wave_min_u32(nxt);
    }
    vlow = key2f(klow);
    vhigh = key2f(khigh);
  }
}

constexpr int HIST_WORDS = 384;                  // 256 bins + dump bin + 64 dump slots, padded to a multiple of 64 words

// Result for ONE pixel from the samples a wave holds (RPL per lane, NaN = no sample, nvalid_lane valid ones in this lane):
// nanmedian (TRIM = false) or the reference's trimmed mean of sorted[t0 : t0 + tn] (TRIM = true).
template <int RPL, bool TRIM>
__device__ __forceinline__ float pixel_result(const float (&val)[RPL], int m, int n, int t0, int tn, unsigned* hist, int lane) {
  float res;
  if (TRIM) {
    int hi_end = t0 + tn;                    // slice [t0, hi_end) of the sorted samples, NaNs (rank >= m) dropped
    if (hi_end > n) hi_end = n;
    if (hi_end > m) hi_end = m;
    if (t0 >= hi_end) {
      res = __uint_as_float(0x7fc00000u);
    } else {
      // the two ends of the slice by the bucket selection (round 5; rounds 1-4: two 32-step bisections on the keys), then one
      // pass for the sum strictly between them and the copies of the end values that fall inside the slice
      float vlo, vhi, dummy;
      rank_vals<RPL>(val, t0, false, hist, lane, vlo, dummy);
      rank_vals<RPL>(val, hi_end - 1, false, hist, lane, vhi, dummy);
      int cle_lo = 0, clt_hi = 0;
      double mid = 0.0;
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        cle_lo += lanes_true(val[r] <= vlo);
        clt_hi += lanes_true(val[r] < vhi);
        if (val[r] > vlo && val[r] < vhi) mid += (double)val[r];
      }
#pragma unroll
      for (int s = 32; s >= 1; s >>= 1) mid += __shfl_xor(mid, s, 64);
      double tot;
      if (vlo == vhi) {
        tot = (double)vlo * (double)(hi_end - t0);
      } else {
        const int nlo = (cle_lo < hi_end ? cle_lo : hi_end) - t0;      // copies of the low value inside the slice
        const int nhi = hi_end - clt_hi;                                 // copies of the high value inside the slice
        tot = mid + (double)vlo * nlo + (double)vhi * nhi;
      }
      res = (float)(tot / (double)(hi_end - t0));
    }
  } else if (m == 0) {
    res = __uint_as_float(0x7fc00000u);
  } else {
    float vlow, vhigh;
    rank_vals<RPL>(val, (m - 1) >> 1, (m & 1) == 0, hist, lane, vlow, vhigh);
    res = (m & 1) ? vlow : (vlow + vhigh) * 0.5f;    // even: (a+b)*0.5 in float32, as numpy
  }
  return res;
}

// TRIM = false: nanmedian.  TRIM = true: mean of sorted[t0 : t0+tn] (np.sort order, NaN last, then nanmean):
// the reference's 'trimmean' (subsampling.py:87-96).
template <int RPL, bool TRIM>
__global__ VIPMI_NO_PK32 __launch_bounds__(512) void median_kernel(const float* __restrict__ cube0, int n, int64_t P,
                                                     int TP, float* __restrict__ out0, int t0, int tn, int ntiles, int xcd_ranges) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 256 histogram words per wave, then the n x (TP+1) tile
  const float* __restrict__ cube = cube0 + (size_t)blockIdx.y * n * P;       // blockIdx.y = cube of the batch
  float* __restrict__ out = out0 + (size_t)blockIdx.y * P;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const int ldt = TP + 1;
  unsigned* hist = reinterpret_cast<unsigned*>(smem) + HIST_WORDS * wave;
  float* tile = smem + HIST_WORDS * nw;
  // workgroup ids go round-robin over the XCDs: XCD x takes a contiguous range of tiles, so neighbouring tiles -- which share
  // their 128-byte lines when a tile is 16 pixels wide -- run on one L2 at about the same time (ntiles = pixel tiles; the grid
  // is rounded up to 8 ranges of equal length)
  // -- C2: 0.319 -> 0.300 ms.  With one workgroup per CU (n = 2000: a 148 KB tile) the plain order is the faster one
  // (9.7 against 10.2 ms): xcd_ranges = 0.
  const int per_xcd = gridDim.x >> 3;
  int64_t tile_id = (int64_t)blockIdx.x;
  if (xcd_ranges == 1) {
    tile_id = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  } else if (xcd_ranges > 1) {
    // chunks of xcd_ranges consecutive tiles dealt round-robin to the XCDs: neighbours still share an L2, but the XCDs no longer
    // own one horizontal band of the image each -- the bands at the top and bottom edge hold the slow pixels of a derotated cube
    // (partly outside the rotated footprint: crowded bins, second levels), and the kernel ended when those two XCDs did
    const int loc = blockIdx.x >> 3;
    tile_id = ((int64_t)(loc / xcd_ranges) * 8 + (blockIdx.x & 7)) * xcd_ranges + loc % xcd_ranges;
  }
  if (tile_id >= ntiles) return;
  const int64_t p0 = tile_id * TP;
  // stage: TP consecutive pixels of every frame.  The kernel is bound by this load (400 row segments of 128 bytes,
  // 1 MB apart), so each thread issues a batch of 16-byte loads (8 threads per segment, 32 frames per pass) before
  // the first LDS write: ~13 requests in flight per thread instead of one.
  if ((TP == 32 || TP == 16) && (P & 3) == 0) {
    const int sh = TP == 32 ? 3 : 2;                                  // TP / 4 threads cover one row segment
    const int seg = threadIdx.x & ((1 << sh) - 1), r0 = threadIdx.x >> sh, RPP = blockDim.x >> sh;   // rows per pass
    const int64_t pc = p0 + 4 * seg;
    const bool inb = pc < P;                                     // P % 4 == 0: the whole float4 is in range
    constexpr int NB = 4;                                        // passes per batch
    for (int fb = 0; fb < n; fb += RPP * NB) {
      float4 v[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int f = fb + r0 + RPP * i;
        v[i] = (inb && f < n) ? *reinterpret_cast<const float4*>(cube + (int64_t)f * P + pc) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int f = fb + r0 + RPP * i;
        if (f < n) {
          float* t = tile + f * ldt + 4 * seg;
          t[0] = v[i].x;
          t[1] = v[i].y;
          t[2] = v[i].z;
          t[3] = v[i].w;
        }
      }
    }
  } else {
    for (int e = threadIdx.x; e < n * TP; e += blockDim.x) {
      const int f = e / TP, j = e % TP;
      const int64_t p = p0 + j;
      tile[f * ldt + j] = (p < P) ? cube[(int64_t)f * P + p] : 0.f;
    }
  }
  block_sync();
  for (int j = wave; j < TP; j += nw) {
    const int64_t p = p0 + j;
    if (p >= P) break;
    float val[RPL];
    int m = 0;
    const float qnan = quiet_nan();
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const int f = lane + 64 * r;
      const bool inside = f < n;                // (a padding lane reads row n - 1 and drops it: no divergent branch)
      const float v = tile[(inside ? f : n - 1) * ldt + j];
      const bool ok = inside && v == v;         // padding and NaN: no sample (every NaN becomes the quiet one)
      val[r] = ok ? v : qnan;
      m += lanes_true(ok);
    }
    const float res = pixel_result<RPL, TRIM>(val, m, n, t0, tn, hist, lane);
    if (lane == 0) out[p] = res;
  }
}

// ---- the same selection with the keys kept in REGISTERS across a chunked staging (1025 .. 2048 frames) ----------------------
// median_kernel stages a whole [n][TP] tile in LDS before the first selection: at n = 2000 that is 136 KB for 16 pixels -- ONE
// workgroup per CU (nothing overlaps its load phase), 64-byte row segments (half of every line fetched is for the neighbour
// tile), 9.6 ms for the 8.4 GB of C5 in the pipeline (0.87 TB/s; 7.3 ms on a cube of normal deviates).  Here a tile is 32 pixels (whole 128-byte lines) and passes through
// LDS in CHUNKS of 64 CH frames (double buffered, 17 / 34 KB each): after every chunk a wave moves the samples of ITS FOUR pixels
// (wave w: pixels w, w + 8, w + 16, w + 24 of the tile) into registers as keys -- 4 x RPL per lane, the full tile never exists
// in LDS.  Workgroups are persistent and walk a flat list of (tile, chunk) items: the global loads of item i + 1 are issued
// before item i is consumed and sit in registers across the selection of a finished tile, so loading and selecting overlap even
// with one workgroup per CU.  Which sample lands in which lane is irrelevant to an order statistic; frames beyond n are staged as
// NaN (key 0xffffffff, sorts last like any NaN).
// (plain functions, not capturing lambdas: a closure that holds the register array by reference is materialised in scratch memory)
// global loads of one (tile, chunk) item of this workgroup: CH rows of 16 bytes per thread
template <int CH, int NCHUNK>
__device__ __forceinline__ void mreg_issue(float4 (&pre)[CH], int item, const float* __restrict__ cube, int n, int64_t P, int seg, int r0,
                                           int vec_ok) {
  const float nanv = __uint_as_float(0x7fc00000u);
  const int64_t tile = (int64_t)blockIdx.x + (int64_t)(item / NCHUNK) * gridDim.x;
  const int f0 = (item % NCHUNK) * (64 * CH);
  const int64_t pc = tile * 32 + 4 * seg;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int f = f0 + r0 + 64 * i;
    float4 v = make_float4(nanv, nanv, nanv, nanv);
    if (f < n) {
      const float* src = cube + (int64_t)f * P + pc;
      if (vec_ok && pc + 3 < P) {
        v = *reinterpret_cast<const float4*>(src);
      } else {
        v.x = pc < P ? src[0] : 0.f;
        v.y = pc + 1 < P ? src[1] : 0.f;
        v.z = pc + 2 < P ? src[2] : 0.f;
        v.w = pc + 3 < P ? src[3] : 0.f;
      }
    }
    pre[i] = v;
  }
}
template <int CH>
__device__ __forceinline__ void mreg_deposit(const float4 (&pre)[CH], float* __restrict__ buf, int seg, int r0) {
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    float* t = buf + (r0 + 64 * i) * 33 + 4 * seg;
    t[0] = pre[i].x;
    t[1] = pre[i].y;
    t[2] = pre[i].z;
    t[3] = pre[i].w;
  }
}

template <int RPL, int CH, bool TRIM>
__global__ VIPMI_NO_PK32 __launch_bounds__(512) void median_reg_kernel(const float* __restrict__ cube0, int n, int64_t P,
                                                                       float* __restrict__ out0, int t0, int tn, int ntiles, int vec_ok) {
  constexpr int TP = 32, LDT = TP + 1, CF = 64 * CH, NCHUNK = RPL / CH, PPW = 4;
  static_assert(RPL % CH == 0, "keys per lane = chunks x 64-frame groups per chunk");
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 8 x HIST_WORDS histogram words, then 2 x [CF][LDT] staging buffers
  const float* __restrict__ cube = cube0 + (size_t)blockIdx.y * n * P;
  float* __restrict__ out = out0 + (size_t)blockIdx.y * P;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned* hist = reinterpret_cast<unsigned*>(smem) + HIST_WORDS * wave;
  float* stage = smem + HIST_WORDS * 8;
  const int seg = threadIdx.x & 7, r0 = threadIdx.x >> 3;         // loader: 8 threads x 16 bytes per row segment, 64 rows per pass
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // tiles blockIdx.x + j gridDim.x
  const int items = my_tiles * NCHUNK;
  float4 pre[CH];
  float key[PPW][RPL];                               // the samples themselves (NaN = none)
  int nvalid[PPW];
  if (items > 0) {
    mreg_issue<CH, NCHUNK>(pre, 0, cube, n, P, seg, r0, vec_ok);
    mreg_deposit<CH>(pre, stage, seg, r0);
  }
  for (int item = 0; item < items; ++item) {
    const int chunk = item % NCHUNK;
    if (item + 1 < items) mreg_issue<CH, NCHUNK>(pre, item + 1, cube, n, P, seg, r0, vec_ok);
    block_sync();                                // the buffer of this item is complete; the other one is free again
    const float* b = stage + (item & 1) * (CF * LDT);
    if (chunk == 0) {
#pragma unroll
      for (int q = 0; q < PPW; ++q) nvalid[q] = 0;
    }
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {              // (compile-time register indices: the chunk number selects the slot)
      if (c == chunk) {
#pragma unroll
        for (int q = 0; q < PPW; ++q)
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            const float v = b[(lane + 64 * i) * LDT + wave + 8 * q];
            const bool ok = v == v;
            key[q][c * CH + i] = ok ? v : quiet_nan();
            nvalid[q] += ok ? 1 : 0;
          }
      }
    }
    if (chunk == NCHUNK - 1) {                      // the tile is complete in registers: select
      const int64_t p0 = ((int64_t)blockIdx.x + (int64_t)(item / NCHUNK) * gridDim.x) * TP;
#pragma unroll
      for (int q = 0; q < PPW; ++q) {
        const int64_t p = p0 + wave + 8 * q;
        if (p < P) {                                // (wave-uniform)
          const float res = pixel_result<RPL, TRIM>(key[q], wave_count<RPL>(nvalid[q]), n, t0, tn, hist, lane);
          if (lane == 0) out[p] = res;
        }
      }
    }
    if (item + 1 < items) mreg_deposit<CH>(pre, stage + ((item + 1) & 1) * (CF * LDT), seg, r0);
  }
}

// out[j][p] = A[j][p] - nanmedian_{i in lib_j} A[i][p]: the "optimised reference" of annular median subtraction
// (psfsub/medsub.py:629-639: for every frame the median of the `nframes` frames closest in time beyond the PA
// threshold).  Libraries are small (nframes, default 4): one thread per (frame, pixel) holds the samples in
// registers and selects by rank counting (NaN-aware, even counts -> mean of the two middle values in float32).
template <int W>
__global__ __launch_bounds__(256) void subset_median_sub_kernel(const float* __restrict__ A, int n, int64_t npx,
                                                                const int32_t* __restrict__ idx,
                                                                const int32_t* __restrict__ len, int wmax,
                                                                float* __restrict__ out) {
  const int j = blockIdx.y;
  const int lj = len[j] < wmax ? len[j] : wmax;
  const int32_t* ij = idx + (size_t)j * wmax;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += (int64_t)gridDim.x * blockDim.x) {
    unsigned key[W];
    int m = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      unsigned kk = 0xffffffffu;
      if (w < lj) {
        const float v = A[(size_t)ij[w] * npx + p];
        if (v == v) {
          kk = f2key(v);
          ++m;
        }
      }
      key[w] = kk;
    }
    float med = __uint_as_float(0x7fc00000u);
    if (m > 0) {
      const int klo = (m - 1) >> 1, khi = m >> 1;
      unsigned vlo = 0, vhi = 0;
#pragma unroll
      for (int a = 0; a < W; ++a) {
        int rank = 0;
#pragma unroll
        for (int b = 0; b < W; ++b) rank += (key[b] < key[a] || (key[b] == key[a] && b < a)) ? 1 : 0;
        if (rank == klo) vlo = key[a];
        if (rank == khi) vhi = key[a];
      }
      const float lo = key2f(vlo), hi = key2f(vhi);
      med = (m & 1) ? lo : (lo + hi) * 0.5f;
    }
    out[(size_t)j * npx + p] = A[(size_t)j * npx + p] - med;
  }
}

__global__ void colreduce_kernel(const float* __restrict__ cube0, int n, int64_t P, int mode,
                                 const float* __restrict__ w, float* __restrict__ out0) {
  const float* __restrict__ cube = cube0 + (size_t)blockIdx.y * n * P;       // blockIdx.y = cube of the batch
  float* __restrict__ out = out0 + (size_t)blockIdx.y * P;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P;
       p += (int64_t)gridDim.x * blockDim.x) {
    double s = 0, s2 = 0;
    float mx = -__builtin_inff();
    int cnt = 0;
    // eight frames per iteration, loads issued before use (a single load in flight per thread left the kernel at a
    // quarter of the HBM rate); frames are still accumulated in index order
    for (int f0 = 0; f0 < n; f0 += 8) {
      float vb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) vb[u] = (f0 + u < n) ? cube[(int64_t)(f0 + u) * P + p] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int f = f0 + u;
        if (f >= n) break;
        const float v = vb[u];
        if (mode == VIPMI_COLLAPSE_STIM) {          // np.mean / np.var semantics: NaN propagates
          s += v;
          s2 += (double)v * v;
          continue;
        }
        if (v == v) {
          ++cnt;
          if (mode == VIPMI_COLLAPSE_ABSMEAN) s += fabsf(v);
          else if (mode == VIPMI_COLLAPSE_WMEAN) s += (double)w[f] * v;
          else s += v;
          mx = fmaxf(mx, v);
        }
      }
    }
    float r;
    const float nanv = __uint_as_float(0x7fc00000u);
    switch (mode) {
      case VIPMI_COLLAPSE_STIM: {
        const double mu = s / n;
        double var = s2 / n - mu * mu;
        if (var < 0) var = 0;
        const float sig = (float)sqrt(var);
        r = (sig != 0.f) ? (float)mu / sig : 0.f;
        if (!(mu == mu)) r = nanv;
        break;
      }
      case VIPMI_COLLAPSE_MEAN:
      case VIPMI_COLLAPSE_ABSMEAN: r = cnt ? (float)(s / cnt) : nanv; break;
      case VIPMI_COLLAPSE_SUM:
      case VIPMI_COLLAPSE_WMEAN: r = (float)s; break;
      default: r = cnt ? mx : nanv; break;
    }
    out[p] = r;
  }
}

// More than 4096 frames (64 keys per lane no longer fit the registers): one THREAD per pixel, the 32-step bitwise
// bisection on the order-preserving keys with the samples re-read from memory in every step (consecutive threads read
// consecutive pixels: coalesced; 34 passes over the cube -- 6000 x 512 x 512 takes ~60 ms, for cubes the eigensolver
// needs 350 ms for).  Same value rules as the register kernel: NaN-aware, even count -> (a + b) * 0.5 in float32.
// rank-k key of a pixel's valid samples, read from memory (32-step bitwise bisection)
__device__ __forceinline__ unsigned stream_select(const float* __restrict__ cube, int n, int64_t P, int64_t p, int k) {
  unsigned ans = 0;
  for (int b = 31; b >= 0; --b) {
    const unsigned cand = ans | (1u << b);
    int c = 0;
    for (int f = 0; f < n; ++f) {
      const float v = cube[(size_t)f * P + p];
      c += (v == v && f2key(v) < cand) ? 1 : 0;
    }
    if (c <= k) ans = cand;
  }
  return ans;
}

// trimmed mean of more than 4096 frames: the two order statistics that bound the slice by streaming bisections, then
// one pass for the sum between them (the value rules of the register kernel: NaNs dropped, ties counted)
__global__ __launch_bounds__(256) void trimmean_stream_kernel(const float* __restrict__ cube0, int n, int64_t P,
                                                              float* __restrict__ out0, int t0, int tn) {
  const float* __restrict__ cube = cube0 + (size_t)blockIdx.y * n * P;
  float* __restrict__ out = out0 + (size_t)blockIdx.y * P;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  int m = 0;
  for (int f = 0; f < n; ++f) {
    const float v = cube[(size_t)f * P + p];
    m += (v == v) ? 1 : 0;
  }
  int hi_end = t0 + tn;
  if (hi_end > n) hi_end = n;
  if (hi_end > m) hi_end = m;
  if (t0 >= hi_end) {
    out[p] = __uint_as_float(0x7fc00000u);
    return;
  }
  const unsigned klo = stream_select(cube, n, P, p, t0), khi = stream_select(cube, n, P, p, hi_end - 1);
  int cle_lo = 0, clt_hi = 0;
  double mid = 0.0;
  for (int f = 0; f < n; ++f) {
    const float v = cube[(size_t)f * P + p];
    if (v == v) {
      const unsigned kf = f2key(v);
      cle_lo += (kf <= klo) ? 1 : 0;
      clt_hi += (kf < khi) ? 1 : 0;
      if (kf > klo && kf < khi) mid += (double)v;
    }
  }
  double tot;
  if (klo == khi) {
    tot = (double)key2f(klo) * (double)(hi_end - t0);
  } else {
    const int nlo = (cle_lo < hi_end ? cle_lo : hi_end) - t0;
    const int nhi = hi_end - clt_hi;
    tot = mid + (double)key2f(klo) * nlo + (double)key2f(khi) * nhi;
  }
  out[p] = (float)(tot / (double)(hi_end - t0));
}

__global__ __launch_bounds__(256) void median_stream_kernel(const float* __restrict__ cube0, int n, int64_t P,
                                                            float* __restrict__ out0) {
  const float* __restrict__ cube = cube0 + (size_t)blockIdx.y * n * P;
  float* __restrict__ out = out0 + (size_t)blockIdx.y * P;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  int m = 0;
  for (int f = 0; f < n; ++f) {
    const float v = cube[(size_t)f * P + p];
    m += (v == v) ? 1 : 0;
  }
  if (m == 0) {
    out[p] = __uint_as_float(0x7fc00000u);
    return;
  }
  const int k = (m - 1) >> 1;
  const unsigned ans = stream_select(cube, n, P, p, k);       // largest key with at most k valid keys below it
  float res = key2f(ans);
  if ((m & 1) == 0) {
    // upper median: the same value again when it is repeated, else the smallest key above it
    int cle = 0;
    unsigned nxt = 0xffffffffu;
    for (int f = 0; f < n; ++f) {
      const float v = cube[(size_t)f * P + p];
      if (v == v) {
        const unsigned kf = f2key(v);
        cle += (kf <= ans) ? 1 : 0;
        if (kf > ans && kf < nxt) nxt = kf;
      }
    }
    const unsigned khigh = (cle >= k + 2) ? ans : nxt;
    res = (key2f(ans) + key2f(khigh)) * 0.5f;
  }
  out[p] = res;
}

template <int RPL, int CH, bool TRIM>
int launch_median_reg(vipmi_ctx* ctx, const float* cube, int64_t batch, int n, int64_t P, float* out, int t0, int tn) {
  constexpr int TP = 32, CF = 64 * CH;
  const size_t lds = (size_t)8 * HIST_WORDS * 4 + (size_t)2 * CF * (TP + 1) * 4;
  auto kern = median_reg_kernel<RPL, CH, TRIM>;
  VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds));
  const int64_t ntiles = cdiv(P, TP);
  // persistent workgroups: as many as are resident (registers allow two per CU up to 8 keys per lane and pixel, the 32-key
  // instance holds 128 key registers: one), a few tiles each
  const int per_cu = RPL <= 8 ? 2 : 1;
  int64_t grid = (int64_t)ctx->num_cu * per_cu;
  if (batch > 1) grid = cdiv(grid, batch);
  if (grid > ntiles) grid = ntiles;
  if (grid < 1) grid = 1;
  const int vec_ok = ((P & 3) == 0 && (reinterpret_cast<uintptr_t>(cube) & 15) == 0) ? 1 : 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)batch), dim3(512), lds, ctx->stream, cube, n, P, out, t0, tn, (int)ntiles, vec_ok);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

template <int RPL, bool TRIM>
int launch_median(vipmi_ctx* ctx, const float* cube, int64_t batch, int n, int64_t P, float* out, int t0, int tn) {
  // pixels per tile: 32 (128-byte row segments) unless 16 lets more workgroups share a CU -- after the bucket-selection
  // rewrite the kernel waits on dependent LDS round trips, so occupancy is what counts (C2, n = 400: 53 + 12 KB -> 2
  // workgroups per CU with 32 pixels, 27 + 12 KB -> 4 with 16: 0.39 -> 0.32 ms)
  int TP = (int)ctx->opt("median_tp", 0);
  if (TP != 16 && TP != 32) {
    const size_t hb = 8 * HIST_WORDS * 4;
    const int wg32 = (int)((160 * 1024) / ((size_t)n * 33 * 4 + hb)), wg16 = (int)((160 * 1024) / ((size_t)n * 17 * 4 + hb));
    TP = (wg32 < 4 && wg16 > wg32) ? 16 : 32;
  }
  const size_t hist_bytes = 8 * HIST_WORDS * 4;             // 8 waves per workgroup
  while (TP > 1 && (size_t)n * (TP + 1) * 4 + hist_bytes > 150 * 1024) TP >>= 1;
  const size_t lds = (size_t)n * (TP + 1) * 4 + hist_bytes;
  auto kern = median_kernel<RPL, TRIM>;
  VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds));
  const int64_t ntiles = cdiv(P, TP);
  // tile -> XCD map: chunks of consecutive tiles dealt round-robin (default: 64 tiles, at least 16 chunks per XCD; option
  // median_xcd_chunk: 1 = one contiguous range per XCD as in rounds 2-4, 0 = plain order).  Derotated residuals of a C2 call:
  // 0.192 ms with ranges, 0.157 with chunks of 64 (tools/median_xcd_ab.py)
  int xr = 0;
  if (lds <= 80 * 1024) {
    xr = (int)ctx->opt("median_xcd_chunk", -1);
    if (xr < 0) {
      xr = (int)(ntiles / 128 < 64 ? ntiles / 128 : 64);
      if (xr < 2) xr = 1;
    }
  }
  const int64_t round_to = xr > 1 ? (int64_t)8 * xr : 8;
  hipLaunchKernelGGL(kern, dim3((unsigned)(cdiv(ntiles, round_to) * round_to), (unsigned)batch), dim3(512), lds, ctx->stream, cube, n, P,
                     TP, out, t0, tn, (int)ntiles, xr);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

}  // namespace

int collapse_f32(vipmi_ctx* ctx, const float* cube, int64_t n, int64_t P, int mode, const float* w,
                 int64_t trim_n, float* out) {
  return collapse_batched_f32(ctx, cube, 1, n, P, mode, w, trim_n, out);
}

// `batch` contiguous cubes [batch][n][P] -> out[batch][P] in one launch (blockIdx.y = cube)
int collapse_batched_f32(vipmi_ctx* ctx, const float* cube, int64_t batch, int64_t n, int64_t P, int mode, const float* w,
                         int64_t trim_n, float* out) {
  VIPMI_REQUIRE(cube && out, "collapse: null pointer");
  VIPMI_REQUIRE(n > 0 && P > 0 && batch > 0 && batch <= 65535, "collapse: bad sizes");
  StageScope sc(ctx, "collapse");
  switch (mode) {
    case VIPMI_COLLAPSE_MEDIAN:
    case VIPMI_COLLAPSE_TRIMMEAN: {
      const bool trim = mode == VIPMI_COLLAPSE_TRIMMEAN;
      int t0 = 0, tn = 0;
      if (trim) {
        // reference: k = (N - n)//2 ; if N%2 != n%2: n += 1 ; mean(sorted[k:k+n])   (subsampling.py:88-96)
        // with python slice semantics for k < 0 or k+n > N (e.g. the default n=50 on a short cube)
        int64_t nn = trim_n;
        VIPMI_REQUIRE(nn > 0, "collapse(trimmean): n must be positive");
        int64_t k = (n - nn) >= 0 ? (n - nn) / 2 : -((nn - n + 1) / 2);   // floor division
        if ((n % 2) != (nn % 2)) nn += 1;
        int64_t e = k + nn;
        if (k < 0) k = (k + n < 0) ? 0 : k + n;
        if (k > n) k = n;
        if (e < 0) e = (e + n < 0) ? 0 : e + n;
        if (e > n) e = n;
        t0 = (int)k;
        tn = (int)(e > k ? e - k : 0);
      }
      const int rpl = (int)cdiv(n, 64);
      // 1025 .. 2048 frames: keys in registers, chunked staging (median_reg_kernel) -- where the tile kernel is down to one workgroup
      // per CU and 64-byte row segments.  Measured (tools/time_median.py, normal deviates + 1 % NaN), tile kernel / register kernel:
      // 2000 x 1024^2 7.26 / 5.87 ms; 1000 x 512^2 0.43 / 0.68, 400 x 512^2 0.18 / 0.29, 200 x 512^2 0.13 / 0.17: with two or more
      // workgroups per CU the tile kernel's 32 waves hide the selection's LDS round trips better than 16 waves with four pixels
      // each, so the smaller instances were dropped again.  Option median_reg = 0 keeps the tile kernel everywhere.
      if (n > 1024 && n <= 2048 && ctx->opt("median_reg", 1) != 0) {
        return trim ? launch_median_reg<32, 4, true>(ctx, cube, batch, (int)n, P, out, t0, tn)
                    : launch_median_reg<32, 4, false>(ctx, cube, batch, (int)n, P, out, 0, 0);
      }
#define VIPMI_MED(R)                                                                              \
  return trim ? launch_median<R, true>(ctx, cube, batch, (int)n, P, out, t0, tn)                 \
              : launch_median<R, false>(ctx, cube, batch, (int)n, P, out, 0, 0)
      if (rpl <= 1) { VIPMI_MED(1); }
      if (rpl <= 2) { VIPMI_MED(2); }
      if (rpl <= 4) { VIPMI_MED(4); }
      if (rpl <= 7) { VIPMI_MED(7); }      // 385..448 frames: one compare less per bisection step (VALU bound)
      if (rpl <= 8) { VIPMI_MED(8); }
      if (rpl <= 16) { VIPMI_MED(16); }
      if (rpl <= 32) { VIPMI_MED(32); }
      if (rpl <= 64) { VIPMI_MED(64); }
#undef VIPMI_MED
      if (!trim)
        hipLaunchKernelGGL(median_stream_kernel, dim3((unsigned)cdiv(P, 256), (unsigned)batch), dim3(256), 0, ctx->stream,
                           cube, (int)n, P, out);
      else
        hipLaunchKernelGGL(trimmean_stream_kernel, dim3((unsigned)cdiv(P, 256), (unsigned)batch), dim3(256), 0, ctx->stream,
                           cube, (int)n, P, out, t0, tn);
      VIPMI_CHECK_HIP(hipGetLastError());
      return VIPMI_OK;
    }
    case VIPMI_COLLAPSE_WMEAN:
      VIPMI_REQUIRE(w != nullptr, "Weights have to be provided for weighted mean mode");
      // fallthrough
    case VIPMI_COLLAPSE_MEAN:
    case VIPMI_COLLAPSE_SUM:
    case VIPMI_COLLAPSE_MAX:
    case VIPMI_COLLAPSE_STIM:
    case VIPMI_COLLAPSE_ABSMEAN: {
      int64_t b = cdiv(P, 256);
      hipLaunchKernelGGL(colreduce_kernel, dim3((unsigned)(b > 8192 ? 8192 : b), (unsigned)batch), dim3(256), 0, ctx->stream,
                         cube, (int)n, P, mode, w, out);
      VIPMI_CHECK_HIP(hipGetLastError());
      return VIPMI_OK;
    }
    default:
      set_error("mode not recognized");
      return VIPMI_ERR_ARG;
  }
}

int subset_median_sub_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* idx,
                          const int32_t* len, int64_t wmax, float* out) {
  VIPMI_REQUIRE(A && idx && len && out, "subset_median_sub: null pointer");
  VIPMI_REQUIRE(n > 0 && npx > 0 && wmax > 0, "subset_median_sub: bad sizes");
  VIPMI_REQUIRE(wmax <= 32, "subset_median_sub: libraries of more than 32 frames are not supported (nframes=%ld)",
                (long)wmax);
  StageScope sc(ctx, "collapse");
  unsigned gx = (unsigned)cdiv(npx, 256);
  if (gx > 1024) gx = 1024;
  const dim3 grid(gx, (unsigned)n), block(256);
#define VIPMI_SMS(W_)                                                                                          \
  hipLaunchKernelGGL(subset_median_sub_kernel<W_>, grid, block, 0, ctx->stream, A, (int)n, npx, idx, len, (int)wmax, out)
  if (wmax <= 4) VIPMI_SMS(4);
  else if (wmax <= 8) VIPMI_SMS(8);
  else if (wmax <= 16) VIPMI_SMS(16);
  else VIPMI_SMS(32);
#undef VIPMI_SMS
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

}  // namespace vipmi
