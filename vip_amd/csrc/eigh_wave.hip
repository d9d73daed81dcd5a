// eigh_wave.hip -- Householder tridiagonalisation of ONE symmetric float64 matrix of 129 .. 448 rows on 64 cooperating
// WAVES (one single-wave workgroup each, all on one XCD), the matrix resident in registers.
//
// Why: the reduction is a chain of ~n dependent steps (reflector -> A v -> exchange -> w -> next reflector); a lone call
// (synchronous mode: the PCA of one cube, psfsub/svd.py:447-470) waits for that chain while the rest of the chip idles.
// tri_multi_kernel (eigh_tri.hip) runs the chain on 16 / 32 workgroups of 16 waves with the rows in LDS: 4.3 us per step at
// n = 400, of which ~3 us are LOCAL -- three workgroup barriers, LDS round trips of the vectors every wave needs, and 16
// waves per CU executing the same float64 vector code on four SIMDs.  Here a participant is a single wave:
//   * row r lives in wave r mod 64 as local row r div 64, element (r, c) in lane c mod 64, register chunk c div 64 --
//     NCH x NCH doubles per lane (7 x 7 = 98 VGPRs at n <= 448); every vector of the step (v, w, p, the next row) is held the
//     same way, REDUNDANTLY per wave: no LDS, no s_barrier, the only reductions are DPP wave sums;
//   * the element of a vector that belongs to local row lr of this wave is (chunk lr, lane = wave id): one v_readlane;
//   * per step ONE exchange through the XCD's L2: every wave publishes beta * (row . v) of its rows with one 8-byte store
//     per row, the owner of row s + 1 publishes that row (= column s + 1 by symmetry) after the pending rank-2 update, then
//     a flag per wave (epoch numbers, one polling lane per flag -- 64 flags = one load per poll), then every wave gathers
//     the two vectors with coalesced 512-byte loads;
//   * work shrinks with the trailing matrix: chunks left of column s + 1 and rows above it are skipped (wave-uniform).
// Placement and visibility follow tri_multi_kernel's one-XCD path (wave_util.h): ids x (mod 8) of an 8 x wider grid, verified
// with HW_REG_XCC_ID behind an agent-scope barrier; agent-scope stores when the check fails.  All 64 waves must be resident
// at once (spin barrier): the launcher uses the kernel for lone synchronous calls only, spins are bounded and latched.
// Output: d, e, tau in det[3][n] and the reflectors in the rows of A (row s, columns > s) -- exactly what stages 2-5 of
// tri_multi_kernel (multisection, inverse iteration, back-transformation, Gram-Schmidt) start from.
#include "common.h"
#include "wave_util.h"
#include "tri_common.h"

namespace vipmi {

namespace {

constexpr int WW = 64;   // participating waves

__device__ __forceinline__ unsigned ld_flag(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// all stores of this wave have landed, publish the epoch, wait for everybody's
__device__ __forceinline__ void wave_exchange(unsigned* flags, unsigned epoch, int wg, bool fast, int* fail) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) {
    if (fast) __hip_atomic_store(flags + wg, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(flags + wg, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned spins = 0;
  bool ok = false;
  while (true) {
    if (!ok) ok = (int)(ld_flag(flags + threadIdx.x) - epoch) >= 0;
    if (__all(ok)) break;
    if (++spins > (1u << 24)) {          // a partner that never became resident must not hang the GPU -- and is reported
      if (fail && threadIdx.x == 0) atomicAdd(fail + 1, 1);
      break;
    }
  }
}

template <int V>
struct IC {
  static constexpr int value = V;
};

// Register arrays only take compile-time indices, while the live part of the matrix starts at column s + 1.  The step loop is
// therefore split by the chunk C0 = (s + 1) / 64 of that column: one copy of the step body per chunk (NCH of them, each over
// the chunks C0 .. NCH-1 and the local rows C0 .. NCH-1 only), inside it everything is straight-line code with static register
// indices and ls = (s + 1) % 64 the only run-time quantity.  (A first version selected the live range with run-time switches:
// 3100 instructions per step, a fifth of them copies at the merge points, 3.6 us per step at n = 400.)
template <int NCH>
__global__ __launch_bounds__(64) void tri_wave_kernel(double* __restrict__ A, int n, double* __restrict__ det,
                                                      double* __restrict__ gb, unsigned* __restrict__ bar, int one_xcd,
                                                      int* __restrict__ fail) {
  static_assert(NCH >= 1 && NCH <= 7, "tri_wave_kernel: up to 448 rows");
  if (one_xcd && (int)(blockIdx.x & 7) != ((one_xcd - 1) & 7)) return;
  const int wg = __builtin_amdgcn_readfirstlane(one_xcd ? blockIdx.x >> 3 : blockIdx.x);
  const int lane = threadIdx.x;
  constexpr int NP = 64 * NCH;
  double* Pb = gb;                 // [2][NP]  beta * (row r . v)
  double* Rb = gb + 2 * NP;        // [2][NP]  row s + 1 after the pending update
  double* Db = gb + 4 * NP;        // last diagonal entry (last step)
  unsigned* xflags = bar + 8;
  unsigned* xids = bar + 72;
  unsigned epoch = 0;

  double rows[NCH][NCH];           // local row lr = global row 64 lr + wg ; chunk ch = columns 64 ch + lane
  double cc[NCH];                  // the row that yields the next reflector
#pragma unroll
  for (int lr = 0; lr < NCH; ++lr) {
    const int r = 64 * lr + wg;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = 64 * ch + lane;
      rows[lr][ch] = (r < n && c < n) ? A[(size_t)r * n + c] : 0.0;
    }
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int c = 64 * ch + lane;
    cc[ch] = (c < n) ? A[c] : 0.0;
  }
  // the exchange buffers are read up to column 64 NCH (+ 1): zero what lies beyond n once (wave 0)
  if (wg == 0) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = 64 * ch + lane;
      if (c >= n) {
        st_shared(&Pb[c], 0.0);
        st_shared(&Pb[NP + c], 0.0);
        st_shared(&Rb[c], 0.0);
        st_shared(&Rb[NP + c], 0.0);
      }
    }
  }
  // everybody has read row 0 and its own rows before a reflector overwrites a row of the input; placement check
  if (lane == 0) __hip_atomic_store(xids + wg, 1u + xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  epoch += 1;
  wave_exchange(xflags, epoch, wg, false, fail);
  const bool fast = one_xcd != 0 && __all(ld_flag(xids + lane) == 1u + xcc_id());
  auto put = [&](double* p, double v) __attribute__((always_inline)) {
    if (fast) st_xcd(p, v);
    else st_shared(p, v);
  };

  double vcur[NCH], vprev[NCH], wprev[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) vprev[ch] = wprev[ch] = vcur[ch] = 0.0;
  double beta = 0.0, v0 = 0.0, e2 = 0.0;
  // Reflector of the row held in cc = row s1 = 64 C0 + ls (chunks C0 ..): x = cc[c > s1]; x0 = x[s1 + 1] and diag = cc[s1] come
  // as scalars.  v = x with v[s1 + 1] = x0 - alpha -> vcur (zero up to column s1), beta = 2 / v.v, and e2 = x[s1 + 2] as a
  // scalar for the next step.
  auto reflect = [&](auto C0c, int ls, double x0, double diag) __attribute__((always_inline)) {
    constexpr int C0 = decltype(C0c)::value;
    const int s1 = 64 * C0 + ls;
    cc[C0] = (lane > ls) ? cc[C0] : 0.0;
    double part = 0.0;
#pragma unroll
    for (int ch = C0; ch < NCH; ++ch) part += cc[ch] * cc[ch];
    const double nrm2 = wave_sum(part);
    const double nrm = nrm2 > 0.0 ? tri::fast_sqrt_pos(nrm2) : 0.0;
    const double alpha = (x0 >= 0.0) ? -nrm : nrm;
    v0 = x0 - alpha;
    double rest = nrm2 - x0 * x0;
    if (rest < 0.0) rest = 0.0;
    const double vv = rest + v0 * v0;
    beta = (nrm2 > 0.0 && vv > 0.0) ? 2.0 * tri::fast_rcp(vv) : 0.0;
#pragma unroll
    for (int ch = C0; ch < NCH; ++ch) vcur[ch] = cc[ch];
    // column s1 + 1 takes v0, column s1 + 2 gives e2: both in chunk C0, or in the next one at its end
    vcur[C0] = (lane == ls + 1) ? v0 : vcur[C0];
    e2 = readlane_f64(cc[C0], (ls + 2) & 63);
    if constexpr (C0 + 1 < NCH) {
      if (ls == 63) vcur[C0 + 1] = (lane == 0) ? v0 : vcur[C0 + 1];
      const double e2n = readlane_f64(cc[C0 + 1], (ls + 2) & 63);
      e2 = (ls + 2 >= 64) ? e2n : e2;
    } else {
      e2 = (ls + 2 >= 64) ? 0.0 : e2;
    }
    if (wg == 0 && lane == 0) {
      det[s1] = diag;
      det[n + s1] = (nrm2 > 0.0) ? alpha : 0.0;
      det[2 * n + s1] = beta;
    }
  };
  reflect(IC<0>{}, 0, readlane_f64(cc[0], 1), readlane_f64(cc[0], 0));

  // one Householder step: column s + 1 = 64 C0 + ls
  auto step = [&](auto C0c, int ls) __attribute__((always_inline)) {
    constexpr int C0 = decltype(C0c)::value;
    const int s = 64 * C0 + ls - 1;
    const int par = s & 1;
    // the owner of row s keeps the reflector for the back-transformation
    if ((s & 63) == wg) {
#pragma unroll
      for (int ch = C0; ch < NCH; ++ch) {
        const int c = 64 * ch + lane;
        if (c > s && c < n) put(&A[(size_t)s * n + c], vcur[ch]);
      }
    }
    // own rows r > s (local rows C0 .., the first one only when wg >= ls), columns from chunk C0 on: pending rank-2 update
    // of step s - 1, then row . v_s
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0;
#pragma unroll
    for (int lr = C0; lr < NCH; ++lr) {
      if (lr > C0 || wg >= ls) {
        const double vr = readlane_f64(vprev[lr], wg), wr = readlane_f64(wprev[lr], wg);
        double a = 0.0;
#pragma unroll
        for (int ch = C0; ch < NCH; ++ch) {
          const double t = rows[lr][ch] - vr * wprev[ch] - wr * vprev[ch];
          rows[lr][ch] = t;
          a += t * vcur[ch];
        }
        acc[lr] = a;
      }
    }
    // row s + 1 (after that update) to everybody
    if (ls == wg) {
#pragma unroll
      for (int ch = C0; ch < NCH; ++ch) {
        const int c = 64 * ch + lane;
        if (c < n) put(&Rb[par * NP + c], rows[C0][ch]);
      }
    }
    if (s + 3 == n && ((n - 1) & 63) == wg) {               // last step: the last diagonal entry as well
      const double dsel = readlane_f64(rows[NCH - 1][NCH - 1], (n - 1) & 63);
      if (lane == 0) put(&Db[0], dsel);
    }
    {
      const double tot = wave_sum8_scatter(acc);            // lane l: row . v of local row l & 7
      const int r = 64 * lane + wg;
      if (lane < NCH && r > s && r < n) put(&Pb[par * NP + r], beta * tot);
    }
    epoch += 1;
    wave_exchange(xflags, epoch, wg, fast, fail);
    // gather beta A v and row s + 1 (left of column s + 1 they are stale: masked), and the entries s + 1, s + 2 of both once
    // more as wave-uniform scalars (vector loads of one address: the scalar cache is not coherent)
    double p[NCH];
    const double p1 = ld_shared(&Pb[par * NP + s + 1]), p2 = ld_shared(&Pb[par * NP + s + 2]);
    const double c1 = ld_shared(&Rb[par * NP + s + 1]), cx = ld_shared(&Rb[par * NP + s + 2]);
#pragma unroll
    for (int ch = C0; ch < NCH; ++ch) {
      p[ch] = ld_shared(&Pb[par * NP + 64 * ch + lane]);
      cc[ch] = ld_shared(&Rb[par * NP + 64 * ch + lane]);
    }
    p[C0] = (lane >= ls) ? p[C0] : 0.0;
    cc[C0] = (lane >= ls) ? cc[C0] : 0.0;
    double kd = 0.0;
#pragma unroll
    for (int ch = C0; ch < NCH; ++ch) kd += vcur[ch] * p[ch];
    const double K = 0.5 * beta * wave_sum(kd);
    const double ws1 = p1 - K * v0;                         // w[s + 1]  (v[s + 1] = v0)
    const double w2 = p2 - K * e2;                          // w[s + 2]  (v[s + 2] = e2)
    const double diag = c1 - 2.0 * v0 * ws1;                // row s + 1 after this step's own update: entries s + 1, s + 2
    const double x0 = cx - v0 * w2 - ws1 * e2;
#pragma unroll
    for (int ch = C0; ch < NCH; ++ch) {
      const double v = vcur[ch];
      const double w = p[ch] - K * v;
      wprev[ch] = w;
      vprev[ch] = v;
      cc[ch] = cc[ch] - v0 * w - ws1 * v;
    }
    if (s + 3 < n) {
      reflect(C0c, ls, x0, diag);
    } else {
      // trailing 2 x 2 block: cc = row n - 2 (fully updated: diag, x0 are its last two entries); the last diagonal entry came
      // through Db and still lacks this step's update  - 2 v[n-1] w[n-1]  (v[n-1] = e2, w[n-1] = w2)
      const double db = ld_shared(&Db[0]) - 2.0 * e2 * w2;
      if (wg == 0 && lane == 0) {
        const int a = n - 2, b = n - 1;
        det[a] = diag;
        det[n + a] = x0;
        det[2 * n + a] = 0.0;
        det[b] = db;
        det[n + b] = 0.0;
        det[2 * n + b] = 0.0;
      }
    }
  };
  auto chunk_steps = [&](auto C0c) __attribute__((always_inline)) {
    constexpr int C0 = decltype(C0c)::value;
    if constexpr (C0 < NCH) {
      const int lo = C0 == 0 ? 1 : 0;                       // s = 64 C0 + ls - 1 >= 0
      int hi = n - 2 - 64 * C0;                             // s <= n - 3
      if (hi > 63) hi = 63;
      for (int ls = lo; ls <= hi; ++ls) step(C0c, ls);
    }
  };
  chunk_steps(IC<0>{});
  chunk_steps(IC<1>{});
  chunk_steps(IC<2>{});
  chunk_steps(IC<3>{});
  chunk_steps(IC<4>{});
  chunk_steps(IC<5>{});
  chunk_steps(IC<6>{});
}

}  // namespace

bool tri_wave_supported(int64_t n) { return n >= 129 && n <= 448; }

// Tridiagonalise A (n x n, symmetric, float64, destroyed): d, e, tau -> det[3][n], reflector s in A[s][s+1 ..].
// bars: 136 zeroed words; gbuf: 4 * 64 * ceil(n / 64) + 8 doubles.  xcd_slot = 1 + XCD to sit on (0: spread, agent scope).
int tri_wave_reduce(vipmi_ctx* ctx, double* A, int n, double* det, double* gbuf, unsigned* bars, int xcd_slot, int* fail) {
  VIPMI_REQUIRE(tri_wave_supported(n), "tri_wave_reduce: unsupported size %d", n);
  const int nch = (int)cdiv(n, 64);
  const dim3 grid(xcd_slot ? 8 * WW : WW), block(64);
  switch (nch) {
    case 3: hipLaunchKernelGGL(tri_wave_kernel<3>, grid, block, 0, ctx->stream, A, n, det, gbuf, bars, xcd_slot, fail); break;
    case 4: hipLaunchKernelGGL(tri_wave_kernel<4>, grid, block, 0, ctx->stream, A, n, det, gbuf, bars, xcd_slot, fail); break;
    case 5: hipLaunchKernelGGL(tri_wave_kernel<5>, grid, block, 0, ctx->stream, A, n, det, gbuf, bars, xcd_slot, fail); break;
    case 6: hipLaunchKernelGGL(tri_wave_kernel<6>, grid, block, 0, ctx->stream, A, n, det, gbuf, bars, xcd_slot, fail); break;
    default: hipLaunchKernelGGL(tri_wave_kernel<7>, grid, block, 0, ctx->stream, A, n, det, gbuf, bars, xcd_slot, fail); break;
  }
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

}  // namespace vipmi
