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
// the opposite sign, as LAPACK dstebz does with its pivmin clamp.
__device__ __forceinline__ int sturm_count(const double* __restrict__ d, const double* __restrict__ e2, int n,
                                           double sigma) {
  double pm = 1.0, p = d[0] - sigma;
  bool neg = p < 0.0 || p == 0.0;          // effective sign of p_i (true = negative); p_0 = 1 is positive
  if (p == 0.0) p = -1e-300;
  int cnt = neg ? 1 : 0;
  for (int i0 = 1; i0 < n; i0 += 16) {
    double db[16], eb[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = i0 + u;
      db[u] = (i < n) ? d[i] : 0.0;
      eb[u] = (i < n) ? e2[i - 1] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (i0 + u < n) {
        const double t = eb[u] * pm;
        double pn = fma(db[u] - sigma, p, -t);
        if (pn == 0.0) pn = neg ? 1e-300 : -1e-300;
        const bool nneg = pn < 0.0;
        cnt += (nneg != neg) ? 1 : 0;
        neg = nneg;
        pm = p;
        p = pn;
      }
    }
    const int ex = ilogb(fabs(p) > fabs(pm) ? p : pm);
    p = scalbn(p, -ex);
    pm = scalbn(pm, -ex);
  }
  return cnt;
}

// One eigenvalue by multisection, executed by a whole wave: the 64 lanes evaluate Sturm counts at 64 interior points
// of the bracket [a, b], which shrinks 65x per sweep.  target = ascending index of the eigenvalue.  Wave-uniform result.
__device__ __forceinline__ double multisect(const double* __restrict__ d, const double* __restrict__ e2, int n,
                                            int target, double a, double b, int lane) {
  for (int sweep = 0; sweep < 14; ++sweep) {
    const double h = (b - a) * (1.0 / 65.0);
    const int cnt = sturm_count(d, e2, n, a + h * (double)(lane + 1));
    const int L = __popcll(__ballot(cnt <= target));          // sigma_l <= lambda_target for the first L lanes
    const double na = (L == 0) ? a : a + h * (double)L;
    const double nb = (L == 64) ? b : a + h * (double)(L + 1);
    a = na;
    b = nb;
    if (b - a <= 2.0 * EPS * fmax(fabs(a), fabs(b)) + 1e-290) break;
  }
  return 0.5 * (a + b);
}

}  // namespace tri
}  // namespace vipmi
