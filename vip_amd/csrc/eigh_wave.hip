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
// Output: d, e, tau in det[3][n] and the reflectors in the rows of A (row s, zeros up to the diagonal) -- what stages 2-5 of
// tri_multi_kernel start from --, plus, from the epilogue, the products of the reflectors inside every group of four and the
// tridiagonal matrix scaled for the multisection.  The file also holds the rest of the lone solve: tri_vec_kernel (eigenvalue,
// inverse iteration and back-transformation of ONE vector per workgroup) and tri_mgs_kernel (Gram-Schmidt, signs); DESIGN 3.2a.
#include "common.h"
#include <mutex>
#include <map>
#include <vector>
#include "wave_util.h"
#include "tri_common.h"

namespace vipmi {

namespace {

constexpr int WW = 64;   // participating waves
constexpr int VG = 4;    // reflectors per group of the back-transformation
constexpr int GW = 8;    // doubles per group in the table of products (VG (VG - 1) / 2 = 6 used)
// position of the product v_a . v_b (a < b < VG) in a group's block
__host__ __device__ constexpr int gram_idx(int a, int b) { return a * (2 * VG - 1 - a) / 2 + (b - a - 1); }

// four full-wave sums, transposed like wave_sum8_scatter (wave_util.h): lane l returns the sum over the wave of x[l & 3]
__device__ __forceinline__ double wave_sum4_scatter(const double (&x)[4]) {
  const int lane = threadIdx.x & 63;
  const bool b0 = lane & 1, b1 = lane & 2;
  double y[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const double keep = b0 ? x[2 * m + 1] : x[2 * m], send = b0 ? x[2 * m] : x[2 * m + 1];
    y[m] = keep + dpp_f64<0xB1>(send);
  }
  const double keep = b1 ? y[1] : y[0], send = b1 ? y[0] : y[1];
  double w = keep + dpp_f64<0x4E>(send);
  w += dpp_f64<0x124>(w);     // row_ror:4
  w += dpp_f64<0x128>(w);     // row_ror:8
  w = swap_add16_f64(w);
  return swap_add32_f64(w);
}

__device__ __forceinline__ unsigned ld_flag(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// all stores of this wave have landed, publish the epoch, wait for everybody's.  Returns false when a partner did not arrive
// within ~1 s (it never became resident: another kernel holds its CU): the failure is latched for vipmi_check_deferred and the
// caller leaves the kernel -- every wave runs into the same time-out at the same epoch, nobody waits for a wave that left.
__device__ __forceinline__ bool wave_exchange(unsigned* flags, unsigned epoch, int wg, bool fast, int* fail) {
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
    if (++spins > (1u << 20)) {
      if (threadIdx.x == 0) {
        if (fail) {
          atomicAdd(fail + 1, 1);
          atomicAdd(flags - 8 + 5, 1u);                                                      // bar[5]: how many of them this launch latched
        }
        __hip_atomic_store(flags - 8 + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // bar[4]: this call is void
      }
      return false;
    }
  }
  return true;
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
                                                      int* __restrict__ fail, double* __restrict__ gram,
                                                      double* __restrict__ det2) {
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
  if (!wave_exchange(xflags, epoch, wg, false, fail)) return;
  const bool fast = one_xcd != 0 && __all(ld_flag(xids + lane) == 1u + xcc_id());
  auto put = [&](double* p, double v) __attribute__((always_inline)) {
    if (fast) st_xcd(p, v);
    else st_shared(p, v);
  };

  double vcur[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) vcur[ch] = 0.0;
  double beta = 0.0, v0 = 0.0, e2 = 0.0;
  bool dead = false;                 // an exchange timed out (wave-uniform): leave
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
    // (the WHOLE row: zeros up to column s, so that the back-transformation reads its rows without masks)
    if ((s & 63) == wg) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int c = 64 * ch + lane;
        if (c < n) put(&A[(size_t)s * n + c], ch >= C0 ? vcur[ch] : 0.0);
      }
    }
    // own rows r > s (local rows C0 .., the first one only when wg >= ls), columns from chunk C0 on: row . v_s  (the rows
    // already carry the rank-2 update of step s - 1: it is applied at the END of that step, see below)
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0;
#pragma unroll
    for (int lr = C0; lr < NCH; ++lr) {
      if (lr > C0 || wg >= ls) {
        double a = 0.0;
#pragma unroll
        for (int ch = C0; ch < NCH; ++ch) a += rows[lr][ch] * vcur[ch];
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
    if (!wave_exchange(xflags, epoch, wg, fast, fail)) {
      dead = true;
      return;
    }
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
    double w[NCH];
#pragma unroll
    for (int ch = C0; ch < NCH; ++ch) {
      const double v = vcur[ch];
      w[ch] = p[ch] - K * v;
      cc[ch] = cc[ch] - v0 * w[ch] - ws1 * v;
    }
    // rank-2 update A <- A - v w^T - w v^T of the own rows r >= s + 2 (row s + 1 lives on as cc), HERE rather than fused into the
    // next step's row pass: these 2/3 of a step's multiply-adds are independent of the reflector's scalar chain below (wave sum,
    // square root, reciprocal) and fill its latency instead of standing between the reflector and the publication of A v
#pragma unroll
    for (int lr = C0; lr < NCH; ++lr) {
      if (lr > C0 || wg > ls) {
        const double vr = readlane_f64(vcur[lr], wg), wr = readlane_f64(w[lr], wg);
#pragma unroll
        for (int ch = C0; ch < NCH; ++ch) rows[lr][ch] = rows[lr][ch] - vr * w[ch] - wr * vcur[ch];
      }
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
      for (int ls = lo; ls <= hi && !dead; ++ls) step(C0c, ls);
    }
  };
  chunk_steps(IC<0>{});
  chunk_steps(IC<1>{});
  chunk_steps(IC<2>{});
  chunk_steps(IC<3>{});
  chunk_steps(IC<4>{});
  chunk_steps(IC<5>{});
  chunk_steps(IC<6>{});

  // Epilogue: the back-transformation (tri_vec_kernel) applies the reflectors in groups of VG with the compact-WY recurrence and
  // needs the products v_a . v_b inside every group -- 64 waves are here with nothing left to do: wave w takes the groups w and
  // w + 64 (group g = reflectors j = n-3 - VG g - q, q = 0 .. VG-1): six products, one transposed wave reduction.  gram[g][GW].
  if (dead) return;
  epoch += 1;
  if (!wave_exchange(xflags, epoch, wg, fast, fail)) return;       // every reflector is in memory / the L2
  const int ngroups = (n - 2 + VG - 1) / VG;
  for (int g = wg; g < ngroups; g += WW) {
    double v[VG][NCH];
#pragma unroll
    for (int q = 0; q < VG; ++q) {
      const int j = n - 3 - VG * g - q;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int c = 64 * ch + lane;
        v[q][ch] = (j >= 0 && c < n) ? ld_shared(&A[(size_t)j * n + c]) : 0.0;
      }
    }
    double part[4] = {0.0, 0.0, 0.0, 0.0}, part2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int a = 0; a < VG; ++a)
#pragma unroll
      for (int b = a + 1; b < VG; ++b) {
        double t = 0.0;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) t += v[a][ch] * v[b][ch];
        if (gram_idx(a, b) < 4) part[gram_idx(a, b)] = t;
        else part2[gram_idx(a, b) - 4] = t;
      }
    const double t0 = wave_sum4_scatter(part), t1 = wave_sum4_scatter(part2);
    if (lane < 4) {
      gram[(size_t)g * GW + lane] = t0;
      gram[(size_t)g * GW + 4 + lane] = t1;
    }
  }
  // ... and wave 0 leaves the tridiagonal matrix the way the multisection wants it: scaled to max-norm 1, the squares of the
  // off-diagonals, the Gershgorin interval:  det2 = { d / scale [n], e / scale [n], (e / scale)^2 [n], scale, lo, hi }
  if (wg == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double dl[NCH], el[NCH], ep[NCH];
    double mx = 0.0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = 64 * ch + lane;
      dl[ch] = (c < n) ? ld_shared(&det[c]) : 0.0;
      el[ch] = (c < n) ? ld_shared(&det[n + c]) : 0.0;
      ep[ch] = (c >= 1 && c < n) ? ld_shared(&det[n + c - 1]) : 0.0;
      mx = fmax(mx, fmax(fabs(dl[ch]), fabs(el[ch])));
    }
    const double scale = wave_max(mx);
    const double iscale = scale > 0.0 ? 1.0 / scale : 0.0;
    double lo = 1e300, hi = -1e300;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = 64 * ch + lane;
      const double d = dl[ch] * iscale, e = el[ch] * iscale, em = ep[ch] * iscale;
      if (c < n) {
        det2[c] = d;
        det2[n + c] = e;
        det2[2 * n + c] = e * e;
        const double rad = fabs(em) + (c + 1 < n ? fabs(e) : 0.0);
        lo = fmin(lo, d - rad);
        hi = fmax(hi, d + rad);
      }
    }
    lo = -wave_max(-lo);
    hi = wave_max(hi);
    const double margin = 4.0 * tri::EPS * (double)n + 1e-290;
    if (lane == 0) {
      det2[3 * n] = scale;
      det2[3 * n + 1] = lo - margin;
      det2[3 * n + 2] = hi + margin;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Stages 2-4 for ONE eigenvector per workgroup (4 waves, one per SIMD), from d, e, tau (det), the reflectors (rows of A) and
// the group products (gram):
//   2. its eigenvalue by multisection: 256 Sturm counts per sweep (bracket / 257: 7 sweeps), the count loop with the zero test
//      off the dependent chain (sturm_count_fast) and the matrix entries as scalar operands (scaled once by tri_wave_kernel);
//   3. inverse iteration (dlagtf-style pivoted LU, two solves) by one lane, the factors in LDS, operands fetched in blocks;
//   4. back-transformation by wave 0: z <- H_0 .. H_{n-3} z, reflectors streamed from L2 one group ahead (registers), applied
//      VG = 4 at a time (two groups of registers, no AGPR traffic) -- the VG products v_q . z come out of ONE transposed wave reduction, the sequential coefficients follow
//      from the group products:  s_q = beta_q (v_q . z - sum_{a<q} s_a v_a . v_q),  z -= sum_q s_q v_q.
// tri_multi_kernel ran these stages on one wave / one lane of a 16-wave workgroup capped at 128 VGPRs: 85 + 155 + 245 us at
// n = 400 (s_memtime, tools/tri_stage_profile.py) -- more than the tridiagonalisation once that ran on eigh_wave's step loop.
// Output: evecs[c][n] not yet orthonormalised against its neighbours (tri_mgs_kernel), evals[c].
template <int NCH>
__global__ __launch_bounds__(256) void tri_vec_kernel(const double* __restrict__ A, int n, const double* __restrict__ det,
                                                      const double* __restrict__ det2, const double* __restrict__ gram,
                                                      double* __restrict__ evals, double* __restrict__ evecs) {
  constexpr int NP = 64 * NCH;
  __shared__ double U0[NP], U1[NP], U2[NP], Lm[NP], Ls[NP], Zl[NP];
  __shared__ double red[2][4];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr double TEPS = tri::EPS;
  // the scaled tridiagonal matrix stays in global memory: its entries are wave-uniform operands (scalar loads, SGPR operands
  // of the recurrences) -- from LDS every step of a Sturm count paid two broadcast reads and their wait
  const double* __restrict__ dd = det2;
  const double* __restrict__ ee = det2 + n;
  const double* __restrict__ e2 = det2 + 2 * n;
  const double scale = det2[3 * n], glo = det2[3 * n + 1], ghi = det2[3 * n + 2];
#ifdef VIPMI_TRI_PROFILE     // stage stamps of workgroup 0 (tools/tri_stage_profile.py): evals[n-16 ..]
  unsigned long long pst[6] = {0, 0, 0, 0, 0, 0};
#define VSTAMP(i) pst[i] = __builtin_amdgcn_s_memtime()
#else
#define VSTAMP(i)
#endif
  VSTAMP(0);
  // ---- 2. multisection ----
  double lam_c;
  {
    const int target = n - 1 - c;
    double a = glo, b = ghi;
    for (int sweep = 0; sweep < 12; ++sweep) {
      const double h = (b - a) * (1.0 / 257.0);
      const int cnt = tri::sturm_count_fast(dd, e2, n, a + h * (double)(tid + 1));
      const int Lw = __popcll(__ballot(cnt <= target));      // sigma <= lambda_target for a prefix of the 256 points
      if (lane == 0) red[sweep & 1][wave] = (double)Lw;
      __syncthreads();
      const int L = (int)(red[sweep & 1][0] + red[sweep & 1][1] + red[sweep & 1][2] + red[sweep & 1][3]);
      const double na_ = (L == 0) ? a : a + h * (double)L;
      const double nb_ = (L == 256) ? b : a + h * (double)(L + 1);
      a = na_;
      b = nb_;
      if (b - a <= 2.0 * TEPS * fmax(fabs(a), fabs(b)) + 1e-290) break;
    }
    lam_c = 0.5 * (a + b);
  }
  VSTAMP(1);
  // ---- 3. inverse iteration (one lane; the start vector and 1 / e are tabulated by everybody first) ----
  // The pass is a chain of n dependent rows on ONE lane, and every instruction costs its 4+ cycles whether 1 or 64 lanes are
  // active: what counts is the instruction count per row.  Rows come in blocks of eight with their operands fetched up front; the
  // pivoting is a (wave-uniform) branch with a handful of instructions on each side -- as selects it was ~75 instructions and ~500
  // cycles per row; the swap branch divides by the sub-diagonal, whose reciprocal is tabulated (IE): no reciprocal on its chain.
  __shared__ double IEs[NP];
  for (int i = tid; i < n; i += 256) {
    Zl[i] = tri::hash_unit((unsigned)i, (unsigned)c);
    IEs[i] = tri::fast_rcp(ee[i]);
  }
  __syncthreads();
  if (tid == 0) {
    const double lc = lam_c - (double)(c + 1) * 4.0 * TEPS;
    const double ptiny = 1e-3 * TEPS;
    double p = dd[0] - lc, q = (n > 1) ? ee[0] : 0.0;
    double yc = Zl[0];
    constexpr int BL = 8;
    auto lu_row = [&](int i, double sub, double nd, double nu, double yn, double isub) __attribute__((always_inline)) {
      double inv, m, u1, u2, yi, sw;
      if (fabs(sub) > fabs(p) && fabs(sub) >= ptiny) {      // interchange: the pivot is the sub-diagonal entry
        inv = isub;
        m = p * inv;
        u1 = nd;
        u2 = nu;
        yi = yn;
        yc = yc - m * yn;
        p = q - m * nd;
        q = -(m * nu);
        sw = 1.0;
      } else {
        if (fabs(p) < ptiny) p = (p < 0.0) ? -ptiny : ptiny;
        inv = tri::fast_rcp(p);
        m = sub * inv;
        u1 = q;
        u2 = 0.0;
        yi = yc;
        yc = yn - m * yc;
        p = nd - m * q;
        q = nu;
        sw = 0.0;
      }
      U0[i] = inv;
      U1[i] = u1;
      U2[i] = u2;
      Lm[i] = m;
      Ls[i] = sw;
      Zl[i] = yi;
    };
    const int nrow = n - 1, nfull = nrow / BL * BL;         // rows 0 .. n-2
    for (int i0 = 0; i0 < nfull; i0 += BL) {
      double sb[BL], db[BL], ub[BL], yb[BL], ib[BL];
#pragma unroll
      for (int u = 0; u < BL; ++u) {                        // (i0 + u + 1 <= n - 1: in range; e[n-1] = 0 is the missing super-diagonal)
        sb[u] = ee[i0 + u];
        db[u] = dd[i0 + u + 1];
        ub[u] = ee[i0 + u + 1];
        yb[u] = Zl[i0 + u + 1];
        ib[u] = IEs[i0 + u];
      }
#pragma unroll
      for (int u = 0; u < BL; ++u) lu_row(i0 + u, sb[u], db[u] - lc, ub[u], yb[u], ib[u]);
    }
    for (int i = nfull; i < nrow; ++i) lu_row(i, ee[i], dd[i + 1] - lc, ee[i + 1], Zl[i + 1], IEs[i]);
    if (fabs(p) < ptiny) p = (p < 0.0) ? -ptiny : ptiny;
    const double invlast = tri::fast_rcp(p);
    double rs = 1.0;
    VSTAMP(4);
    for (int it = 0; it < 2; ++it) {
      if (it > 0) {
        yc = Zl[0] * rs;
        auto fw_row = [&](int i, double zn, double m, double sw) __attribute__((always_inline)) {
          const double yn = zn * rs;
          double yi;
          if (sw != 0.0) {
            yi = yn;
            yc = yc - m * yn;
          } else {
            yi = yc;
            yc = yn - m * yc;
          }
          Zl[i] = yi;
        };
        for (int i0 = 0; i0 < nfull; i0 += BL) {
          double zb[BL], mb[BL], lb[BL];
#pragma unroll
          for (int u = 0; u < BL; ++u) {
            zb[u] = Zl[i0 + u + 1];
            mb[u] = Lm[i0 + u];
            lb[u] = Ls[i0 + u];
          }
#pragma unroll
          for (int u = 0; u < BL; ++u) fw_row(i0 + u, zb[u], mb[u], lb[u]);
        }
        for (int i = nfull; i < nrow; ++i) fw_row(i, Zl[i + 1], Lm[i], Ls[i]);
        VSTAMP(5);
      }
      double x1 = yc * invlast, x2 = 0.0;
      Zl[n - 1] = x1;
      double acc = x1 * x1;
      auto bw_row = [&](int i, double z, double a0, double a1, double a2) __attribute__((always_inline)) {
        const double x = (z - a1 * x1 - a2 * x2) * a0;
        Zl[i] = x;
        acc += x * x;
        x2 = x1;
        x1 = x;
      };
      int i = n - 2;
      for (; i >= BL - 1; i -= BL) {
        double zb[BL], a0[BL], a1[BL], a2[BL];
#pragma unroll
        for (int u = 0; u < BL; ++u) {
          zb[u] = Zl[i - u];
          a0[u] = U0[i - u];
          a1[u] = U1[i - u];
          a2[u] = U2[i - u];
        }
#pragma unroll
        for (int u = 0; u < BL; ++u) bw_row(i - u, zb[u], a0[u], a1[u], a2[u]);
      }
      for (; i >= 0; --i) bw_row(i, Zl[i], U0[i], U1[i], U2[i]);
      rs = acc > 0.0 ? 1.0 / sqrt(acc) : 1.0;
    }
    red[0][0] = rs;                                         // normalisation of the final solution
  }
  __syncthreads();
  VSTAMP(2);
  if (wave != 0) return;
  // ---- 4. back-transformation (wave 0) ----
  const double rs = red[0][0];
  double z[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int i = 64 * ch + lane;
    z[ch] = (i < n) ? Zl[i] * rs : 0.0;
  }
  const int ngroups = (n - 2 + VG - 1) / VG;
  const double* tau = det + 2 * n;
  // rows are clean (zeros up to the diagonal: tri_wave_kernel) -- no masks; the last chunk is read clamped into the row, what
  // the lanes beyond n pick up there meets z = 0 in the products and is removed from z after the update
  const int last = min(64 * (NCH - 1) + lane, n - 1);
  const bool tail = 64 * (NCH - 1) + lane < n;
  auto fetch = [&](int g, double (&v)[VG][NCH]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < VG; ++q) {
      const double* row = A + (size_t)max(n - 3 - VG * g - q, 0) * n;
#pragma unroll
      for (int ch = 0; ch + 1 < NCH; ++ch) v[q][ch] = row[64 * ch + lane];
      v[q][NCH - 1] = row[last];
    }
  };
  auto apply = [&](int g, const double (&v)[VG][NCH]) __attribute__((always_inline)) {
    const int j0 = n - 3 - VG * g;
    double y[4], sc[VG];
#pragma unroll
    for (int q = 0; q < VG; ++q) {
      double t = 0.0;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) t += v[q][ch] * z[ch];
      y[q] = t;
    }
    const double tot = wave_sum4_scatter(y);                  // lane l: v_(l & 3) . z
    // the group's six products and four betas: two contiguous scalar loads, issued with the row loads (fetched one at a time
    // where they are used, their latency sat four times in the recurrence's chain)
    double Gv[6], tv[VG];
    {
      const double* G = gram + (size_t)g * GW;
#pragma unroll
      for (int i = 0; i < 6; ++i) Gv[i] = G[i];
      const int jb = max(j0 - (VG - 1), 0);                   // betas of reflectors jb .. jb + 3 (the last group may start at 0)
#pragma unroll
      for (int i = 0; i < VG; ++i) tv[i] = tau[jb + i];
#pragma unroll
      for (int q = 0; q < VG; ++q) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < VG; ++i) t = (j0 - q - jb == i) ? tv[i] : t;
        sc[q] = (j0 - q >= 0) ? t : 0.0;                     // (sc[q] holds beta_q until the recurrence overwrites it)
      }
    }
#pragma unroll
    for (int q = 0; q < VG; ++q) {
      double d = readlane_f64(tot, q);
#pragma unroll
      for (int a = 0; a < q; ++a) d -= sc[a] * Gv[gram_idx(a, q)];
      sc[q] = sc[q] * d;
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      double t = z[ch];
#pragma unroll
      for (int q = 0; q < VG; ++q) t -= sc[q] * v[q][ch];
      z[ch] = t;
    }
    z[NCH - 1] = tail ? z[NCH - 1] : 0.0;
  };
  double va[VG][NCH], vb[VG][NCH];
  fetch(0, va);
  for (int g = 0; g < ngroups; g += 2) {
    if (g + 1 < ngroups) fetch(g + 1, vb);
    apply(g, va);
    if (g + 2 < ngroups) fetch(g + 2, va);
    if (g + 1 < ngroups) apply(g + 1, vb);
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int i = 64 * ch + lane;
    if (i < n) evecs[(size_t)c * n + i] = z[ch];
  }
  if (lane == 0) evals[c] = lam_c * scale;
#ifdef VIPMI_TRI_PROFILE
  VSTAMP(3);
  if (c == 0 && lane == 0)
    for (int i = 0; i < 6; ++i) evals[n - 16 + i] = (double)(pst[i] - pst[0]);
#endif
}

// 5. modified Gram-Schmidt over the k vectors in eigenvalue order (inverse iteration leaves the vectors of close eigenvalues
// nearly parallel), sign convention (largest-magnitude component positive).  One workgroup: vector c in wave c mod 16, slot
// c div 16 (registers), the current vector broadcast through LDS.
template <int NCH>
__global__ __launch_bounds__(1024) void tri_mgs_kernel(int n, int k, double* __restrict__ evals, double* __restrict__ evecs,
                                                       const unsigned* __restrict__ bar) {
  constexpr int NP = 64 * NCH, NW = 16, VPW = 4;
  __shared__ double qv[NP];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = k < n ? k : n;
  if (bar[4] != 0u) {
    // the reduction timed out (its 64 waves were not co-resident: another kernel held their CUs for a second): the failure is
    // latched for vipmi_check_deferred, and a synchronous caller that never asks must not get plausible numbers -- NaN
    const double bad = __longlong_as_double(0x7ff8000000000000ll);
    for (int e = tid; e < k * n; e += 1024) evecs[e] = bad;
    for (int e = tid; e < k; e += 1024) evals[e] = bad;
    return;
  }
  double zz[VPW][NCH];
#pragma unroll
  for (int v = 0; v < VPW; ++v) {
    const int c = wave + NW * v;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int i = 64 * ch + lane;
      zz[v][ch] = (c < kk && i < n) ? evecs[(size_t)c * n + i] : 0.0;
    }
  }
  for (int c = 0; c < kk; ++c) {
    const int ow = c % NW, ov = c / NW;
    if (wave == ow) {
#pragma unroll
      for (int v = 0; v < VPW; ++v)
        if (v == ov) {
          double sq = 0.0;
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch) sq += zz[v][ch] * zz[v][ch];
          sq = wave_sum(sq);
          const double inv = sq > 0.0 ? 1.0 / sqrt(sq) : 0.0;
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch) {
            zz[v][ch] *= inv;
            qv[64 * ch + lane] = zz[v][ch];
          }
        }
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < VPW; ++v) {
      const int c2 = wave + NW * v;
      if (c2 > c && c2 < kk) {
        double sq = 0.0;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) sq += zz[v][ch] * qv[64 * ch + lane];
        sq = wave_sum(sq);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) zz[v][ch] -= sq * qv[64 * ch + lane];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int v = 0; v < VPW; ++v) {
    const int c = wave + NW * v;
    if (c < k) {
      double best = -1.0, bval = 0.0;
      int bidx = 0x7fffffff;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int i = 64 * ch + lane;
        const double a = fabs(zz[v][ch]);
        if (i < n && a > best) {
          best = a;
          bval = zz[v][ch];
          bidx = i;
        }
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        const double ob = __shfl_xor(best, m, 64), ovv = __shfl_xor(bval, m, 64);
        const int oi = __shfl_xor(bidx, m, 64);
        if (ob > best || (ob == best && oi < bidx)) {
          best = ob;
          bval = ovv;
          bidx = oi;
        }
      }
      const double sg = (c < kk) ? (bval < 0.0 ? -1.0 : 1.0) : 0.0;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int i = 64 * ch + lane;
        if (i < n) evecs[(size_t)c * n + i] = zz[v][ch] * sg;
      }
      if (lane == 0 && c >= kk) evals[c] = 0.0;
    }
  }
}

}  // namespace

// ---- launches whose workgroups wait for each other, ordered per device (common.h: CoopOrder) ----
namespace {
std::mutex coop_mu;
std::map<int, std::pair<std::vector<hipEvent_t>, int>> coop_ring;
}  // namespace

CoopOrder::CoopOrder(vipmi_ctx* c) : ctx(c), status(VIPMI_OK), locked(false) {
  coop_mu.lock();
  locked = true;
  auto& ring = coop_ring[ctx->device];
  if (ring.first.empty()) {
    ring.first.resize(8);
    for (auto& e : ring.first)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) status = VIPMI_ERR_HIP;
    ring.second = -1;
  }
  if (status == VIPMI_OK && ring.second >= 0 && hipStreamWaitEvent(ctx->stream, ring.first[ring.second], 0) != hipSuccess)
    status = VIPMI_ERR_HIP;
}
int CoopOrder::done() {
  auto& ring = coop_ring[ctx->device];
  int st = status;
  if (st == VIPMI_OK) {
    ring.second = (ring.second + 1) % (int)ring.first.size();
    if (hipEventRecord(ring.first[ring.second], ctx->stream) != hipSuccess) st = VIPMI_ERR_HIP;
  }
  if (locked) coop_mu.unlock();
  locked = false;
  return st;
}
CoopOrder::~CoopOrder() {
  if (locked) coop_mu.unlock();
}

bool tri_wave_supported(int64_t n) { return n >= 129 && n <= 448; }

// The reduction spins on 64 single-wave workgroups of ONE XCD (num_cu / 8 CUs): on a part with few CUs per XCD, or with a register
// budget per wave that lets fewer of them share a SIMD, they cannot all be resident and every call would run into its time-out.
// Asked once per (device, chunk count) from the occupancy calculator; the callers fall back to the multi-workgroup kernel.
bool tri_wave_fits(vipmi_ctx* ctx, int64_t n) {
  if (!tri_wave_supported(n) || ctx->num_cu % 8 != 0) return false;
  static std::mutex mu;
  static std::map<std::pair<int, int>, bool> seen;
  const int nch = (int)cdiv(n, 64);
  std::lock_guard<std::mutex> lk(mu);
  auto it = seen.find({ctx->device, nch});
  if (it != seen.end()) return it->second;
  int per_cu = 0;
  hipError_t e;
  switch (nch) {
    case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tri_wave_kernel<3>, 64, 0); break;
    case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tri_wave_kernel<4>, 64, 0); break;
    case 5: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tri_wave_kernel<5>, 64, 0); break;
    case 6: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tri_wave_kernel<6>, 64, 0); break;
    default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tri_wave_kernel<7>, 64, 0); break;
  }
  const bool ok = e == hipSuccess && (int64_t)per_cu * (ctx->num_cu / 8) >= WW;
  seen[{ctx->device, nch}] = ok;
  return ok;
}

// Tridiagonalise A (n x n, symmetric, float64, destroyed): d, e, tau -> det[3][n], reflector s in A[s][s+1 ..], the products of the
// reflector groups of four -> gram[ceil((n-2)/4)][8], the scaled tridiagonal matrix and its Gershgorin interval -> det2[3 n + 3].  bars: 136 zeroed words; gbuf: 4 * 64 * ceil(n / 64) + 8 doubles.
// xcd_slot = 1 + XCD to sit on (0: spread, agent scope).
int tri_wave_reduce(vipmi_ctx* ctx, double* A, int n, double* det, double* gbuf, unsigned* bars, int xcd_slot, int* fail, double* gram,
                    double* det2) {
  VIPMI_REQUIRE(tri_wave_supported(n), "tri_wave_reduce: unsupported size %d", n);
  const int nch = (int)cdiv(n, 64);
  // (test hook, option eigh_wave_drop = 1: launch one participant short -- every exchange then runs into its time-out, which is
  //  how tests/test_gpu_kernels.py exercises the failure path: NaN results, latched error, the next call unaffected)
  const int drop = ctx->opt("eigh_wave_drop", 0) != 0 ? 1 : 0;
  const dim3 grid(xcd_slot ? 8 * (WW - drop) : WW - drop), block(64);
  // One wave kernel at a time per device (process-wide): its 64 waves spin on each other, and two launches that became resident
  // only in part at the same moment (several host threads, each on its own stream) could fill an XCD and wait for waves that no
  // longer fit -- until the time-out.  Every launch waits for the event the previous one recorded; a lone caller never waits.
  CoopOrder order(ctx);
  VIPMI_TRY(order.status);
  switch (nch) {
    case 3: hipLaunchKernelGGL(tri_wave_kernel<3>, grid, block, 0, ctx->stream, A, n, det, gbuf, bars, xcd_slot, fail, gram, det2); break;
    case 4: hipLaunchKernelGGL(tri_wave_kernel<4>, grid, block, 0, ctx->stream, A, n, det, gbuf, bars, xcd_slot, fail, gram, det2); break;
    case 5: hipLaunchKernelGGL(tri_wave_kernel<5>, grid, block, 0, ctx->stream, A, n, det, gbuf, bars, xcd_slot, fail, gram, det2); break;
    case 6: hipLaunchKernelGGL(tri_wave_kernel<6>, grid, block, 0, ctx->stream, A, n, det, gbuf, bars, xcd_slot, fail, gram, det2); break;
    default: hipLaunchKernelGGL(tri_wave_kernel<7>, grid, block, 0, ctx->stream, A, n, det, gbuf, bars, xcd_slot, fail, gram, det2); break;
  }
  VIPMI_CHECK_HIP(hipGetLastError());
  return order.done();
}

// Stages 2-5 after tri_wave_reduce: the leading k eigenpairs (k <= 64), one workgroup per vector, then the Gram-Schmidt pass.
int tri_wave_vectors(vipmi_ctx* ctx, const double* A, int n, int k, const double* det, const double* det2, const double* gram,
                     double* evals, double* evecs, const unsigned* bars) {
  VIPMI_REQUIRE(tri_wave_supported(n) && k >= 1 && k <= 64, "tri_wave_vectors: unsupported sizes n=%d k=%d", n, k);
  const int nch = (int)cdiv(n, 64), kk = k < n ? k : n;
  switch (nch) {
    case 3:
      hipLaunchKernelGGL(tri_vec_kernel<3>, dim3(kk), dim3(256), 0, ctx->stream, A, n, det, det2, gram, evals, evecs);
      hipLaunchKernelGGL(tri_mgs_kernel<3>, dim3(1), dim3(1024), 0, ctx->stream, n, k, evals, evecs, bars);
      break;
    case 4:
      hipLaunchKernelGGL(tri_vec_kernel<4>, dim3(kk), dim3(256), 0, ctx->stream, A, n, det, det2, gram, evals, evecs);
      hipLaunchKernelGGL(tri_mgs_kernel<4>, dim3(1), dim3(1024), 0, ctx->stream, n, k, evals, evecs, bars);
      break;
    case 5:
      hipLaunchKernelGGL(tri_vec_kernel<5>, dim3(kk), dim3(256), 0, ctx->stream, A, n, det, det2, gram, evals, evecs);
      hipLaunchKernelGGL(tri_mgs_kernel<5>, dim3(1), dim3(1024), 0, ctx->stream, n, k, evals, evecs, bars);
      break;
    case 6:
      hipLaunchKernelGGL(tri_vec_kernel<6>, dim3(kk), dim3(256), 0, ctx->stream, A, n, det, det2, gram, evals, evecs);
      hipLaunchKernelGGL(tri_mgs_kernel<6>, dim3(1), dim3(1024), 0, ctx->stream, n, k, evals, evecs, bars);
      break;
    default:
      hipLaunchKernelGGL(tri_vec_kernel<7>, dim3(kk), dim3(256), 0, ctx->stream, A, n, det, det2, gram, evals, evecs);
      hipLaunchKernelGGL(tri_mgs_kernel<7>, dim3(1), dim3(1024), 0, ctx->stream, n, k, evals, evecs, bars);
      break;
  }
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

}  // namespace vipmi
