// tri_common.h -- device helpers shared by the tridiagonal eigensolvers (eigh_tri.hip, eigh_tri_large.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace vipmi {
namespace tri {

constexpr double EPS = 2.220446049250313e-16;

// 1/x to float64 accuracy from the hardware seed and two Newton steps (a dependent chain of 5 operations instead of
// the ~15 of an IEEE division; used inside the pivoted tridiagonal LU, where the chain length is the run time)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  r = r * (2.0 - x * r);
  return r;
}

// sqrt(x) for x > 0 to float64 accuracy from the hardware reciprocal-square-root seed and two Newton steps (a chain of
// ~8 operations instead of the ~25 of the IEEE sequence with its scaling and fix-up; used where the chain is the run time)
__device__ __forceinline__ double fast_sqrt_pos(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * (1.5 - 0.5 * x * r * r);
  r = r * (1.5 - 0.5 * x * r * r);
  const double s = x * r;
  return s + 0.5 * r * (x - s * s);             // one correction of the root itself
}

// deterministic pseudo-random start vector entry in (-1, 1)
__device__ __forceinline__ double hash_unit(unsigned a, unsigned b) {
  unsigned x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
  x ^= x >> 15;
  x *= 0x2C1B3C6Du;
  x ^= x >> 12;
  x *= 0x297A2D39u;
  x ^= x >> 15;
  return ((double)(x >> 8) + 0.5) * (2.0 / 16777216.0) - 1.0;
}

// Number of eigenvalues of the (scaled, max-norm 1) tridiagonal (d, e2 = e^2) strictly below sigma: sign changes of
// the Sturm sequence p_i = (d_i - sigma) p_{i-1} - e_{i-1}^2 p_{i-2} (one dependent FMA per step instead of a
// float64 division); the pair (p_i, p_{i-1}) is renormalised every 16 steps (|growth| <= 5 per step), the operands
// of 16 steps are fetched from LDS up front (uniform addresses).  A zero term counts as a sign change and is given
// the opposite sign, as LAPACK dstebz does with its pivmin clamp.  The signs of a block are shifted into an integer
// (one v_alignbit per step) and the changes counted once per block: the multisection is bound by the instruction
// count of this loop (9 per step; 17 with a compare / select / add per step).
__device__ __forceinline__ unsigned hi_word(double x) { return (unsigned)(__double_as_longlong(x) >> 32); }

__device__ __forceinline__ double sturm_step(double dmi, double e2i, double sigma, double& pm, double p, unsigned& sg) {
  const double t = e2i * pm;
  double pn = fma(dmi - sigma, p, -t);
  // +-1e-300 with the sign opposite to p (p is never zero: zeros are replaced as they appear)
  const unsigned alt_hi = 0x01A56E1Fu | (~hi_word(p) & 0x80000000u);
  const double alt = __longlong_as_double(((long long)alt_hi << 32) | 0xC2F8F359ll);
  pn = (pn == 0.0) ? alt : pn;
  sg = __builtin_amdgcn_alignbit(sg, hi_word(pn), 31);      // (sg << 1) | sign(pn)
  pm = p;
  return pn;
}

// SQ: `e2` holds the off-diagonals themselves and is squared as it is fetched (the solver for more than 2048 rows has
// no LDS left for the squares; the multiplications are off the dependent chain).
template <bool SQ = false>
__device__ __forceinline__ int sturm_count(const double* __restrict__ d, const double* __restrict__ e2, int n,
                                           double sigma) {
  double pm = 1.0, p = d[0] - sigma;
  if (p == 0.0) p = -1e-300;               // p_{-1} = 1 is positive
  unsigned sg = hi_word(p) >> 31;          // bit 0 = sign of the newest term
  int cnt = (int)sg;
  int i0 = 1;
  for (; i0 + 16 <= n; i0 += 16) {
    double db[16], eb[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      db[u] = d[i0 + u];
      eb[u] = e2[i0 + u - 1];
      if (SQ) eb[u] *= eb[u];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) p = sturm_step(db[u], eb[u], sigma, pm, p, sg);
    cnt += __popc((sg ^ (sg >> 1)) & 0xffffu);
    const int ex = ilogb(fabs(p) > fabs(pm) ? p : pm);
    p = scalbn(p, -ex);
    pm = scalbn(pm, -ex);
  }
  const int m = n - i0;                    // 0 .. 15 remaining steps
  for (int u = 0; u < m; ++u) {
    const double eo = e2[i0 + u - 1];
    p = sturm_step(d[i0 + u], SQ ? eo * eo : eo, sigma, pm, p, sg);
  }
  cnt += __popc((sg ^ (sg >> 1)) & ((1u << m) - 1u));
  return cnt;
}

// The same count with the zero test OFF the dependent chain: sturm_step's compare / select sits between two FMAs of the
// recurrence (~55 cycles per step measured, 85 us for the 20 leading values of a 400-row problem).  Here a block of 16 steps runs
// on plain FMAs while a flag collects `some term was exactly zero`; only then (never, in practice) the block is redone with the
// careful step.  Same counts as sturm_count, bit for bit.
__device__ __forceinline__ int sturm_count_fast(const double* __restrict__ d, const double* __restrict__ e2, int n, double sigma) {
  double pm = 1.0, p = d[0] - sigma;
  if (p == 0.0) p = -1e-300;
  unsigned sg = hi_word(p) >> 31;
  int cnt = (int)sg;
  int i0 = 1;
  for (; i0 + 16 <= n; i0 += 16) {
    double db[16], eb[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      db[u] = d[i0 + u] - sigma;
      eb[u] = e2[i0 + u - 1];
    }
    double fp = p, fpm = pm;
    unsigned fsg = sg;
    bool zero = false;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const double pn = fma(db[u], fp, -(eb[u] * fpm));
      zero = zero || (pn == 0.0);
      fsg = __builtin_amdgcn_alignbit(fsg, hi_word(pn), 31);
      fpm = fp;
      fp = pn;
    }
    if (__builtin_expect(__any(zero), 0)) {                 // redo the block with the careful step (wave-uniform branch)
#pragma unroll
      for (int u = 0; u < 16; ++u) p = sturm_step(d[i0 + u], eb[u], sigma, pm, p, sg);
    } else {
      p = fp;
      pm = fpm;
      sg = fsg;
    }
    cnt += __popc((sg ^ (sg >> 1)) & 0xffffu);
    const int ex = ilogb(fabs(p) > fabs(pm) ? p : pm);
    p = scalbn(p, -ex);
    pm = scalbn(pm, -ex);
  }
  const int m = n - i0;
  for (int u = 0; u < m; ++u) p = sturm_step(d[i0 + u], e2[i0 + u - 1], sigma, pm, p, sg);
  cnt += __popc((sg ^ (sg >> 1)) & ((1u << m) - 1u));
  return cnt;
}

// One eigenvalue by multisection, executed by a whole wave: the 64 lanes evaluate Sturm counts at 64 interior points
// of the bracket [a, b], which shrinks 65x per sweep.  target = ascending index of the eigenvalue.  Wave-uniform result.
template <bool SQ = false>
__device__ __forceinline__ double multisect(const double* __restrict__ d, const double* __restrict__ e2, int n,
                                            int target, double a, double b, int lane) {
  for (int sweep = 0; sweep < 14; ++sweep) {
    const double h = (b - a) * (1.0 / 65.0);
    const int cnt = sturm_count<SQ>(d, e2, n, a + h * (double)(lane + 1));
    const int L = __popcll(__ballot(cnt <= target));          // sigma_l <= lambda_target for the first L lanes
    const double na = (L == 0) ? a : a + h * (double)L;
    const double nb = (L == 64) ? b : a + h * (double)(L + 1);
    a = na;
    b = nb;
    if (b - a <= 2.0 * EPS * fmax(fabs(a), fabs(b)) + 1e-290) break;
  }
  return 0.5 * (a + b);
}

// The kk largest eigenvalues by ONE wave at once (kk <= 32): the 64 lanes are split into kk segments of P = 64 / kk interior points,
// segment e works on the bracket of the e-th largest eigenvalue, which shrinks (P + 1)x per sweep.  One wave per eigenvalue
// (multisect) gains log2(65) = 6 bits per sweep and eigenvalue; this gains kk log2(P + 1) bits per sweep -- 28 bits for kk = 10 --, so
// the kk values cost ~20 sweeps of one wave instead of 9 kk: where the solver is bound by the number of vector instructions it
// issues (the second launch of big batches: six workgroups per CU, 98 % VALU busy, half of it Sturm counts) that is what counts.
// out[e] (e < kk): the e-th largest eigenvalue, written by the first lane of its segment.  n1 = ascending index of the largest.
template <bool SQ = false>
__device__ __forceinline__ void multisect_many(const double* __restrict__ d, const double* __restrict__ e2, int n, int kk, double a0,
                                               double b0, int lane, double* __restrict__ out) {
  const int P = 64 / kk;
  const int seg = lane / P, j = lane - seg * P;
  const bool active = seg < kk;
  const int target = n - 1 - (active ? seg : 0);
  const unsigned long long segmask = (P >= 64) ? ~0ull : ((1ull << P) - 1ull);
  const int shift = (active ? seg : 0) * P;
  const double ip = 1.0 / (double)(P + 1);
  double a = a0, b = b0;
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double h = (b - a) * ip;
    const int cnt = sturm_count<SQ>(d, e2, n, a + h * (double)(j + 1));
    const unsigned long long m = __ballot(cnt <= target);       // inside a segment: sigma <= lambda_target for its first L lanes
    const int L = __popcll((m >> shift) & segmask);
    const double na = (L == 0) ? a : a + h * (double)L;
    const double nb = (L == P) ? b : a + h * (double)(L + 1);
    a = na;
    b = nb;
    const bool done = !active || (b - a <= 2.0 * EPS * fmax(fabs(a), fabs(b)) + 1e-290);
    if (__all(done)) break;
  }
  if (active && j == 0) out[seg] = 0.5 * (a + b);
}

}  // namespace tri
}  // namespace vipmi
