// eigh_tri.hip -- top-k eigenpairs of small symmetric float64 matrices (n <= 512, k <= 64), batched: one
// workgroup per problem, the classic dense path  A = Q T Q^T  ->  eig(T)  ->  back-transformation:
//
//   1. Householder tridiagonalisation (unblocked, right-looking).  The rank-2 update of step s-1 and the
//      matrix-vector product of step s are fused into ONE pass over the trailing matrix (row s is updated first
//      and yields the next Householder vector), so the matrix is read and written once per step.  The matrix
//      lives in global memory (L2-resident: 400^2 float64 = 1.3 MB); vectors v, w, p, d, e in LDS.
//   2. top-k eigenvalues of T by multisection: one wave per eigenvalue, 64 Sturm counts (one per lane) shrink
//      the bracket 65x per sweep -- 10 sweeps instead of 53 bisection steps.
//   3. eigenvectors of T by inverse iteration (one lane per vector, pivoted tridiagonal LU as LAPACK dlagtf,
//      3 iterations), then modified Gram-Schmidt over the k vectors (one wave per vector, registers).
//   4. back-transformation Z <- H_0 H_1 ... H_{n-3} Z, one wave per vector, no workgroup barrier.
//
// The PCA only needs the k leading eigenpairs of the Gram matrix (svd_wrapper(..., ncomp), psfsub/svd.py:342-620;
// get_eigenvectors, svd.py:623-702); the one-sided Jacobi solver of eigh.hip computes all n of them through
// ~5000 dependent rotation steps (10.6 ms at n = 400), this path needs ~n dependent steps.  Same output contract
// as eigh.hip for the leading k rows: evals descending, evecs[c*n + i] unit norm with the largest-magnitude
// component positive.  Zero-padded problems (annular PCA: library sizes differ per frame) pass their active size.
#include "common.h"
#include <atomic>
#include <memory>
#include <unistd.h>
#include "wave_util.h"
#include "tri_common.h"

namespace vipmi {

namespace {

constexpr int TNT = 1024;           // threads per workgroup
constexpr int TNW = TNT / 64;       // waves
constexpr double TEPS = tri::EPS;
using tri::fast_rcp;
using tri::hash_unit;
using tri::sturm_count;

// NT = threads per workgroup: 1024 for a handful of problems; 512 once there are more problems than CUs (annular
// PCA: 400 problems of 200 x 200 per annulus), so that two problems share a CU -- a problem is bound by the latency
// of its ~n dependent steps (7.6 us each at n = 200), not by throughput: 400 problems take 2.7 ms, one takes 1.5 ms.
// The Gram-Schmidt stage holds 4 vectors per wave: k <= NT/16.
// RPW > 0: the matrix lives in REGISTERS during the tridiagonalisation -- wave w holds rows w, w + TNW, ... (RPW of them,
// cyclic, so every wave keeps work as the trailing matrix shrinks), lane l the columns l, l + 64, ... (RPL of them) of
// each: RPW * RPL doubles per thread (200 x 200 on 512 threads: 100).  Round 1 streamed the trailing triangle from L2
// in every step: 400 problems of 200 x 200 in flight moved ~20 TB/s, the aggregate L2 bandwidth, which is what the 7.6 us
// per step of the batched solver were (annular PCA: 24 of C3's 35 ms).  The whole (symmetric) square is kept, so A v
// needs only column sums -- per lane, in registers, no wave reduction -- combined across the waves through LDS.
// workgroup barrier that orders LDS traffic only (no wait for outstanding global stores)
constexpr int TRI_BAR_WORDS = 136;            // per problem: counter (8 words), 64 flags, 64 XCC ids (tri_multi_kernel)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// (256 threads: capped at 80 VGPRs so that six workgroups share a CU -- that variant runs the latency-bound second half of
// split batches; 3200 problems 7.10 -> 6.83 ms against four workgroups per CU, eight are slower again)
// Optional input gather of the register-resident variant: problem p reads its matrix as G[p / per_seg][idx[p][r]][idx[p][c]]
// (the zero-padded library sub-Gram matrix of annular PCA, straight from the segment's Gram matrix) instead of A[p][r][c];
// A[p] is then only the workspace that receives the reflectors.  G == nullptr: no gather.
struct TriGather {
  const double* G = nullptr;
  const int32_t* idx = nullptr;
  int ldg = 0, stride = 0, per_seg = 1;
};

template <int RPL, int NT, int RPW = 0>
__global__ __launch_bounds__(NT, (NT == 256 ? 6 : 1)) void tri_eig_kernel(double* __restrict__ Aall, int n, int k,
                                                     const int32_t* __restrict__ nact, double* __restrict__ evals_all,
                                                     double* __restrict__ evecs_all, double* __restrict__ scratch_all,
                                                     int kp, int all_evals, int kc, int phase = 0,
                                                     double* __restrict__ det_all = nullptr,
                                                     const unsigned* __restrict__ guard = nullptr,
                                                     const unsigned* __restrict__ gcount = nullptr,
                                                     int* __restrict__ fail = nullptr, TriGather gat = TriGather()) {
  // guard != nullptr: this launch is the RECOVERY of a cooperating solve (launch_tri_multi): it follows every such solve in the
  // stream, returns at once when that solve went through (*guard == 0) and otherwise -- a partner of the cooperating kernel never
  // became resident within its time-out: another process held the CUs -- solves the problem(s) again from a copy of the input, alone
  // on one CU each (no workgroup of this kernel waits for another).  The time-outs the dead launch latched (*gcount of them in
  // fail[1]) are taken back and fail[3] counts the recovery: the caller gets the right eigenpairs and a counter instead of an error.
  if (guard != nullptr && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
  const bool multisect_many_on = (phase & 4) != 0;     // (phase + 4: the leading eigenvalues of T by one wave at once)
  phase &= 3;
  // phase: 0 = the whole solve; 1 = tridiagonalisation only (d, e, tau -> det_all[prob][3][n], reflectors in A);
  // 2 = everything after it from those arrays.  Big batches run the two halves as two launches: the second half is a
  // chain of latencies on a few waves, so it runs with 256 threads and four problems per CU while the register-
  // resident first half (one problem per CU) already works on the next libraries.
  constexpr int TNT = NT, TNW = NT / 64;          // (shadow the file-scope values used by tri_multi_kernel)
  extern __shared__ double sm[];
  double* vcur = sm;             // [n] Householder vector of the current step (indexed by absolute row)
  double* vprev = vcur + n;      // [n] pending rank-2 update  A -= v w^T + w v^T
  double* wprev = vprev + n;     // [n]
  double* pcur = wprev + n;      // [n]
  double* dd = pcur + n;         // [n] diagonal of T
  double* ee = dd + n;           // [n] ee[i] couples i and i+1
  double* tau = ee + n;          // [n] Householder scalars
  double* e2 = tau + n;          // [n] (scaled) squared off-diagonals
  double* lam = e2 + n;          // [64] scaled eigenvalues, descending
  const int prob = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* A = Aall + (size_t)prob * n * n;
  double* evals = evals_all + (size_t)prob * n;
  double* evecs = evecs_all + (size_t)prob * n * n;
  double* scr = scratch_all + (size_t)prob * 6 * n * kp;
  int na = nact ? nact[prob] : n;
  if (na > n) na = n;
  const int kk = k < na ? k : na;
  double* det = det_all ? det_all + (size_t)prob * 3 * n : nullptr;
  for (int i = tid; i < n; i += TNT) {
    vprev[i] = 0.0;
    wprev[i] = 0.0;
    vcur[i] = 0.0;
  }
  __syncthreads();

#ifdef VIPMI_TRI_PROFILE
#define TRI_STAMP(i) do { __syncthreads(); if (prob == 0 && tid == 0) evals[n - 8 + (i)] = (double)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define TRI_STAMP(i)
#endif
  TRI_STAMP(0);
  if (phase == 2) {
    for (int i = tid; i < n; i += TNT) {
      dd[i] = det[i];
      ee[i] = det[n + i];
      tau[i] = det[2 * n + i];
    }
  } else {
  // ---------------- 1. tridiagonalisation ----------------
  // Per step: ONE global round trip (the trailing pass) over the LOWER triangle of the trailing matrix only (the
  // batch is bound by the traffic of these passes: 400 problems of 200 x 200 move 2/3 n^3 * 8 B each).  Element
  // (r, c), c <= r, is updated once and used twice: in the dot product of row r with v (wave reduction) and, for
  // c < r, in the column sum  sum_r A[r][c] v[r]  that a lane accumulates in registers for its columns; the column
  // sums of the waves are combined through LDS in a fixed order (deterministic).  The first batch of loads is
  // issued before the Householder vector of the step is formed; column s+1 of the updated matrix (the next
  // Householder column) is handed over through LDS (`nrow`) instead of being re-read.
  if constexpr (RPW > 0) {
    double* pcolw = lam + 72 + n;                      // [TNW][n] per-wave column parts of A v (same place as below)
    // RPL = 4 serves 129..200 rows: a fourth register chunk would hold 8 live columns in 64 lanes (25 wasted doubles per
    // thread: the kernel spilled inside the step loop).  By symmetry the entries (r, c >= 192) of rows r < 192 are the
    // entries (c, r) of rows 192..199, which the waves hold anyway (row 192 + w is row block RPW-1 of wave w): the column
    // sums (A v)[c >= 192] become ONE row dot product per wave, and only the 8 x 8 corner needs storage of its own
    // (`ablk`: lanes 0..7 of wave w hold A[192 + w][192 + lane]).
    constexpr int CH = (RPL == 4) ? 3 : RPL;
    constexpr bool TAIL8 = (RPL == 4);
    // the three vectors the row blocks read, padded to TNW RPW entries and zero outside the live rows (rows <= s of the
    // current Householder vector, rows >= na of all three): the row blocks need no per-row masks
    constexpr int NP = TNW * RPW;
    double* const vprev_n = vprev;                     // the n-strided arrays: the closing formulas read them
    double* const wprev_n = wprev;
    double* vcur = pcolw + TNW * n;                    // (shadow the n-strided arrays above)
    double* vprev = vcur + NP;
    double* wprev = vprev + NP;
    double* nrow = wprev + NP;                         // [NP] column s of the updated matrix (the next Householder column)
    double* xch = nrow + NP;                           // [16] per-wave partial sums and broadcast values of the vector phases
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // in an SGPR: row conditions are scalar branches
    for (int i = tid; i < 4 * NP; i += TNT) vcur[i] = 0.0;
    __syncthreads();
    double areg[RPW][CH];
    // gathered input (annular PCA): the library's rows / columns of the segment's Gram matrix -- 1.3 MB per segment, L2-resident;
    // the sub-Gram matrices are never written out and read back (1 GB each way at C3)
    const double* Gp = gat.G ? gat.G + (size_t)(prob / gat.per_seg) * gat.ldg * gat.ldg : nullptr;
    const int32_t* ip = gat.G ? gat.idx + (size_t)prob * gat.stride : nullptr;
    int gc[CH], gc8 = 0;
    if (Gp) {
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) gc[ch] = (lane + 64 * ch < na) ? ip[lane + 64 * ch] : 0;
      if constexpr (TAIL8) gc8 = (lane < 8 && 64 * CH + lane < na) ? ip[64 * CH + lane] : 0;
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int r = wave + TNW * j;
      const double* src = Gp ? Gp + (size_t)(r < na ? ip[r] : 0) * gat.ldg : A + (size_t)r * n;
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) {
        const int c = lane + 64 * ch;
        areg[j][ch] = (r < na && c < na) ? src[Gp ? gc[ch] : c] : 0.0;
      }
    }
    double ablk = 0.0;
    if constexpr (TAIL8) {
      const int r8 = 64 * CH + wave, c8 = 64 * CH + lane;
      const bool ok8 = lane < 8 && r8 < na && c8 < na;
      if (Gp) ablk = ok8 ? Gp[(size_t)ip[r8 < na ? r8 : 0] * gat.ldg + gc8] : 0.0;
      else ablk = ok8 ? A[(size_t)r8 * n + c8] : 0.0;
      // (A v)[192 + w] is written by wave w alone, into the slot of wave 0: the other waves' slots stay zero
      for (int e = tid; e < TNW * 8; e += TNT)
        if (64 * CH + (e & 7) < n) pcolw[(e >> 3) * n + 64 * CH + (e & 7)] = 0.0;
    }
    if (Gp) {
      const size_t g0 = na > 0 ? (size_t)ip[0] : 0;
      for (int c = tid; c < na; c += TNT) nrow[c] = Gp[(size_t)ip[c] * gat.ldg + g0];   // column 0
      if (na == 1 && tid == 0) A[0] = Gp[g0 * gat.ldg + g0];                           // (the closing formulas read it from A)
    } else {
      for (int c = tid; c < na; c += TNT) nrow[c] = A[(size_t)c * n];  // column 0
    }
    __syncthreads();
#ifdef VIPMI_TRI_PROFILE
    long long seg[5] = {0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define SEG(i) do { const long long tn = __builtin_amdgcn_s_memtime(); seg[i] += tn - tlast; tlast = tn; } while (0)
#else
#define SEG(i)
#endif
    // The vector phases of a step are spread over the first NV = ceil(n / 64) waves (one element per lane: element
    // r = base + 64 wave + lane), partial sums and the few broadcast values go through `xch`.  Reflector of step s1 from
    // the (updated) column s1 (x = its element r = s1 + 64 wave + lane): called by waves < NV between two barriers.
    constexpr int NV = RPL;                               // waves that take part (RPL * 64 >= n)
    // phase A (before a barrier): partial norms and the two leading entries; phase B (after it): the vector itself
    auto reflector_a = [&](int s1, double x) {
      const int r = s1 + 64 * wave + lane;
      const double part = wave_sum((r > s1 && r < na) ? x * x : 0.0);
      if (lane == 0) xch[8 + wave] = part;
      if (wave == 0 && lane == 0) xch[12] = x;            // diagonal entry
      if (wave == 0 && lane == 1) xch[13] = x;            // first entry below it
    };
    auto reflector_b = [&](int s1, double x) {
      double nrm2 = xch[8];
#pragma unroll
      for (int w = 1; w < NV; ++w) nrm2 += xch[8 + w];
      const double xd = xch[12], x0 = xch[13];
      // (square root and reciprocal from the hardware seeds + Newton; tiny or huge norms take the IEEE sequence)
      const bool easy = nrm2 > 1e-280 && nrm2 < 1e280;
      const double nrm = easy ? tri::fast_sqrt_pos(nrm2) : sqrt(nrm2);
      const double alpha = (x0 >= 0.0) ? -nrm : nrm;
      const double v0 = x0 - alpha;
      double rest = nrm2 - x0 * x0;
      if (rest < 0.0) rest = 0.0;
      const double vv = rest + v0 * v0;
      const double beta = (nrm2 > 0.0 && vv > 0.0) ? (easy ? 2.0 * tri::fast_rcp(vv) : 2.0 / vv) : 0.0;
      const int r = s1 + 64 * wave + lane;
      if (r < na) vcur[r] = (r == s1) ? 0.0 : ((r == s1 + 1) ? v0 : x);
      if (wave == 0 && lane == 0) {
        dd[s1] = xd;
        ee[s1] = (nrm2 > 0.0) ? alpha : 0.0;
        tau[s1] = beta;
      }
    };
    {
      const int r = 64 * wave + lane;
      const double x = (wave < NV && r < na) ? nrow[r] : 0.0;
      if (wave < NV && na > 2) reflector_a(0, x);
      lds_barrier();
      if (wave < NV && na > 2) reflector_b(0, x);
      lds_barrier();
    }
    // Four barriers per step: (1) all waves: pending rank-2 update fused with the column sums of A v, capture of row s+1;
    // (2) waves < NV, one element per lane: p = beta A v from the per-wave sums (all loads of an element issued together),
    // partial v.p; (3) K, w = p - K v, column s+1 with its own step's update, partial norms; (4) the next reflector.
    for (int s = 0; s + 2 < na; ++s) {
      const double beta = tau[s];
      // kept for the back-transform, in the (otherwise unused) upper triangle: row s, columns > s.  The only global
      // access of a step: the barriers inside the step loop wait for LDS only (a full __syncthreads waits for the write
      // acknowledgement from L2, ~1.5 us per step on the critical path)
      for (int c = s + 1 + tid; c < na; c += TNT) A[(size_t)s * n + c] = vcur[c];
      // the lane's columns: pending update vectors and this step's Householder vector
      double vpc[CH], wpc[CH], colacc[CH];
      bool live[CH];
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) {
        const int c = lane + 64 * ch;
        live[ch] = c > s && c < na;
        vpc[ch] = live[ch] ? vprev[c] : 0.0;
        wpc[ch] = live[ch] ? wprev[c] : 0.0;
        colacc[ch] = 0.0;
      }
      double vcc[CH], v8 = 0.0, w8 = 0.0, vc8 = 0.0;  // TAIL8: this step's vector at the lane's columns; the three vectors
      if constexpr (TAIL8) {                          // at column 192 + lane (lanes 0..7)
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) vcc[ch] = vcur[lane + 64 * ch];
        if (lane < 8) {
          v8 = vprev[64 * CH + lane];
          w8 = wprev[64 * CH + lane];
          vc8 = vcur[64 * CH + lane];
        }
      }
      // rank-2 update of step s-1 fused with the column sums of A v for step s: rows r > s of this wave, in groups of GR
      // rows without branches inside a group (the 3 GR LDS reads of a group are in flight together; a branch per row
      // serialised 25 LDS round trips per step).  Rows r <= s of a live group take a meaningless update (they are never
      // read again) and add nothing to the column sums (vcur is zero there); rows >= na stay zero.
      constexpr int GR = (RPW % 5 == 0) ? 5 : 4;
      const int jcap = (wave == ((s + 1) & (TNW - 1))) ? (s + 1) / TNW : -1;     // this wave's row block that is row s+1
#pragma unroll
      for (int g = 0; g < RPW / GR; ++g) {
        if (wave + TNW * (g * GR + GR - 1) > s && wave + TNW * g * GR < na) {      // (wave-uniform)
          double vr[GR], wr[GR], vcr[GR];
#pragma unroll
          for (int i = 0; i < GR; ++i) {
            const int r = wave + TNW * (g * GR + i);
            vr[i] = vprev[r];
            wr[i] = wprev[r];
            vcr[i] = vcur[r];
          }
#pragma unroll
          for (int i = 0; i < GR; ++i) {
            const int j = g * GR + i;
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
              const double t = areg[j][ch] - vr[i] * wpc[ch] - wr[i] * vpc[ch];
              areg[j][ch] = t;
              colacc[ch] = fma(t, vcr[i], colacc[ch]);
            }
          }
          // row s+1 (= column s+1 by symmetry) without the still pending update of this step: the next Householder
          // column (entries of dead columns are never read)
          if (jcap >= g * GR && jcap < g * GR + GR) {
#pragma unroll
            for (int i = 0; i < GR; ++i)
              if (jcap == g * GR + i) {
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) nrow[lane + 64 * ch] = areg[g * GR + i][ch];
              }
          }
        }
      }
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) {
        const int c = lane + 64 * ch;
        if (live[ch]) pcolw[wave * n + c] = colacc[ch];
      }
      if constexpr (TAIL8) {
        const int r8 = 64 * CH + wave;
        if (r8 > s && r8 < na) {                       // (wave-uniform)
          ablk = ablk - vprev[r8] * w8 - wprev[r8] * v8;
          double rd = ablk * vc8;
#pragma unroll
          for (int ch = 0; ch < CH; ++ch) rd = fma(areg[RPW - 1][ch], vcc[ch], rd);
          rd = wave_sum(rd);
          if (lane == 0) pcolw[r8] = rd;
          // entry r8 of the next Householder column: A[r8][s+1] (for r8 = s+1 the diagonal entry)
          const int cs = s + 1;
          double val = ablk;
          if (cs < 64) val = areg[RPW - 1][0];
          else if (cs < 128) val = areg[RPW - 1][1];
          else if (cs < 192) val = areg[RPW - 1][2];
          if (lane == (cs < 192 ? (cs & 63) : cs - 192)) nrow[r8] = val;
        }
      }
      SEG(0);                                    // profile build: update + corner
      lds_barrier();
      SEG(1);                                    // wait at the barrier
      double pv = 0.0, vl = 0.0, xl = 0.0;
      const int rv = s + 1 + 64 * wave + lane;   // this lane's element in the vector phases
      const bool okv = wave < NV && rv < na;
      if (wave < NV) {
        const int rr = okv ? rv : na - 1;
        double q[TNW];
#pragma unroll
        for (int w = 0; w < TNW; ++w) q[w] = pcolw[w * n + rr];
        const double v_ = vcur[rr], x_ = nrow[rr];
        double t = q[0];
#pragma unroll
        for (int w = 1; w < TNW; ++w) t += q[w];
        pv = okv ? beta * t : 0.0;
        vl = okv ? v_ : 0.0;
        xl = okv ? x_ : 0.0;
        const double part = wave_sum(vl * pv);
        if (lane == 0) xch[wave] = part;
        if (wave == 0 && lane == 0) {
          xch[4] = vl;                             // v[s+1], p[s+1]
          xch[5] = pv;
        }
      }
      lds_barrier();
      if (wave < NV) {
        double kd = xch[0];
#pragma unroll
        for (int w = 1; w < NV; ++w) kd += xch[w];
        const double K = 0.5 * beta * kd;
        const double vs1 = xch[4], ws1 = xch[5] - K * vs1;
        const double w = pv - K * vl;
        if (okv) {
          wprev[rv] = w;
          vprev[rv] = vl;
        }
        xl = xl - vs1 * w - ws1 * vl;              // column s+1 with its own step's update
        if (s + 3 < na) reflector_a(s + 1, xl);
      }
      SEG(2);                                    // vector phases
      lds_barrier();
      if (wave < NV && s + 3 < na) reflector_b(s + 1, xl);
      lds_barrier();
      SEG(3);
    }
#ifdef VIPMI_TRI_PROFILE
    if (prob == 0 && tid == 0) for (int i = 0; i < 5; ++i) evals[n - 16 + i] = (double)seg[i];
#endif
    // the trailing 2 x 2 block (without the last pending update) goes back to memory for the closing formulas below,
    // the pending update vectors to where those formulas read them
    for (int i = tid; i < na; i += TNT) {
      vprev_n[i] = vprev[i];
      wprev_n[i] = wprev[i];
    }
    // the last 2x2 block goes back through LDS (global addresses of all RPW rows would stay live across the step loop)
    double* fin = pcolw;
    if (na >= 2) {
#pragma unroll
      for (int j = 0; j < RPW; ++j)
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
          const int r = wave + TNW * j, c = lane + 64 * ch;
          if (r >= na - 2 && r < na && c >= na - 2 && c < na) fin[(r - (na - 2)) * 2 + (c - (na - 2))] = areg[j][ch];
        }
      if constexpr (TAIL8) {
        const int r8 = 64 * CH + wave, c8 = 64 * CH + lane;
        if (lane < 8 && r8 >= na - 2 && r8 < na && c8 >= na - 2 && c8 < na) fin[(r8 - (na - 2)) * 2 + (c8 - (na - 2))] = ablk;
      }
    }
    __syncthreads();
    if (na >= 2 && tid < 4) A[(size_t)(na - 2 + (tid >> 1)) * n + (na - 2 + (tid & 1))] = fin[tid];
    __syncthreads();
  } else {
  constexpr int RQ = 4;                                // rows per wave and batch
  double* nrow = e2;                                   // [n] column s of the updated matrix (e2 is not yet in use)
  double* prow = lam + 72;                             // [n] row parts of A v
  double* pcolw = prow + n;                            // [TNW][n] per-wave column parts of A v
  for (int c = tid; c < na; c += TNT) nrow[c] = A[(size_t)c * n];  // column 0
  __syncthreads();
  for (int s = 0; s + 2 < na; ++s) {
    // first batch of the trailing pass: rows r0 .. r0+RQ-1 of this wave (independent of the new Householder vector)
    double a[RQ][RPL];
    const int rfirst = s + 1 + RQ * wave;
#pragma unroll
    for (int q = 0; q < RQ; ++q)
#pragma unroll
      for (int ch = 0; ch < RPL; ++ch) {
        const int c = s + 1 + lane + 64 * ch, r = rfirst + q;
        a[q][ch] = (r < na && c <= r) ? A[(size_t)r * n + c] : 0.0;
      }
    if (wave == 0) {
      // column s below the diagonal (already updated) is the next Householder vector
      double nrm2 = 0.0;
      for (int c = s + 1 + lane; c < na; c += 64) {
        const double x = nrow[c];
        vcur[c] = x;
        nrm2 += x * x;
      }
      nrm2 = wave_sum(nrm2);
      const double x0 = nrow[s + 1];
      const double nrm = sqrt(nrm2);
      const double alpha = (x0 >= 0.0) ? -nrm : nrm;
      const double v0 = x0 - alpha;
      double rest = nrm2 - x0 * x0;
      if (rest < 0.0) rest = 0.0;
      const double vv = rest + v0 * v0;
      const double beta = (nrm2 > 0.0 && vv > 0.0) ? 2.0 / vv : 0.0;
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) {
        dd[s] = nrow[s];
        vcur[s + 1] = v0;
        ee[s] = (nrm2 > 0.0) ? alpha : 0.0;
        tau[s] = beta;
      }
      __builtin_amdgcn_wave_barrier();
      // kept for the back-transform, in the (otherwise unused) upper triangle: row s, columns > s
      for (int c = s + 1 + lane; c < na; c += 64) A[(size_t)s * n + c] = vcur[c];
    }
    __syncthreads();
    const double beta = tau[s];
    double colacc[RPL];
#pragma unroll
    for (int ch = 0; ch < RPL; ++ch) colacc[ch] = 0.0;
    for (int r0 = rfirst; r0 < na; r0 += RQ * TNW) {
      if (r0 != rfirst) {
#pragma unroll
        for (int q = 0; q < RQ; ++q)
#pragma unroll
          for (int ch = 0; ch < RPL; ++ch) {
            const int c = s + 1 + lane + 64 * ch, r = r0 + q;
            a[q][ch] = (r < na && c <= r) ? A[(size_t)r * n + c] : 0.0;
          }
      }
#pragma unroll
      for (int q = 0; q < RQ; ++q) {
        const int r = r0 + q;
        if (r < na) {
          const double vr = vprev[r], wr = wprev[r], vcr = vcur[r];
          double acc = 0.0;
#pragma unroll
          for (int ch = 0; ch < RPL; ++ch) {
            const int c = s + 1 + lane + 64 * ch;
            if (c <= r) {
              const double t = a[q][ch] - vr * wprev[c] - wr * vprev[c];
              A[(size_t)r * n + c] = t;
              acc += t * vcur[c];
              if (c < r) colacc[ch] += t * vcr;
              if (ch == 0 && lane == 0) nrow[r] = t;     // column s+1 (without the still pending update of this step)
            }
          }
          acc = wave_sum(acc);
          if (lane == 0) prow[r] = acc;
        }
      }
    }
#pragma unroll
    for (int ch = 0; ch < RPL; ++ch) {
      const int c = s + 1 + lane + 64 * ch;
      if (c < na) pcolw[wave * n + c] = colacc[ch];
    }
    __syncthreads();
    for (int r = s + 1 + tid; r < na; r += TNT) {
      double t = prow[r];
#pragma unroll 4
      for (int w = 0; w < TNW; ++w) t += pcolw[w * n + r];
      pcur[r] = beta * t;
    }
    __syncthreads();
    // K = beta/2 v.p (every wave computes it: no further barrier) ; w = p - K v becomes the pending update
    double kd = 0.0;
    for (int r = s + 1 + lane; r < na; r += 64) kd += vcur[r] * pcur[r];
    const double K = 0.5 * beta * wave_sum(kd);
    const double vs1 = vcur[s + 1], ws1 = pcur[s + 1] - K * vs1;
    for (int r = s + 1 + tid; r < na; r += TNT) {
      const double v = vcur[r];
      const double w = pcur[r] - K * v;
      wprev[r] = w;
      vprev[r] = v;
      nrow[r] = nrow[r] - vs1 * w - ws1 * v;      // column s+1 with its own step's update: ready for the next step
    }
    __syncthreads();
  }
  }
  if (tid == 0) {
    if (na >= 2) {
      const int a = na - 2, b = na - 1;
      dd[a] = A[(size_t)a * n + a] - 2.0 * vprev[a] * wprev[a];
      ee[a] = A[(size_t)b * n + a] - vprev[a] * wprev[b] - wprev[a] * vprev[b];
      dd[b] = A[(size_t)b * n + b] - 2.0 * vprev[b] * wprev[b];
      ee[b] = 0.0;
    } else if (na == 1) {
      dd[0] = A[0];
      ee[0] = 0.0;
    }
  }
  }   // phase != 2
  __syncthreads();
  if (phase == 1) {
    for (int i = tid; i < n; i += TNT) {
      det[i] = dd[i];
      det[n + i] = ee[i];
      det[2 * n + i] = tau[i];
    }
    return;
  }

  TRI_STAMP(1);
  // ---------------- 2. leading eigenvalues of T (scaled to max-norm 1) ----------------
  double scale = 0.0, glo = 0.0, ghi = 0.0;
  {
    double mx = 0.0;
    for (int i = lane; i < na; i += 64) mx = fmax(mx, fmax(fabs(dd[i]), fabs(ee[i])));
    scale = wave_max(mx);
  }
  const double iscale = scale > 0.0 ? 1.0 / scale : 0.0;
  __syncthreads();
  for (int i = tid; i < na; i += TNT) {
    const double e = ee[i] * iscale;
    dd[i] *= iscale;
    ee[i] = e;
    e2[i] = e * e;
  }
  __syncthreads();
  {
    double lo = 1e300, hi = -1e300;
    for (int i = lane; i < na; i += 64) {
      const double rad = (i > 0 ? fabs(ee[i - 1]) : 0.0) + (i + 1 < na ? fabs(ee[i]) : 0.0);
      lo = fmin(lo, dd[i] - rad);
      hi = fmax(hi, dd[i] + rad);
    }
    glo = -wave_max(-lo);
    ghi = wave_max(hi);
    const double margin = 4.0 * TEPS * (double)na + 1e-290;
    glo -= margin;
    ghi += margin;
  }
  // throughput mode (the 256-thread second launch of big batches: issue-bound): all kk values in one wave at once
  // (from 8 values on: below that one wave per value is as cheap and finishes in fewer sweeps -- 1100 problems of 200 rows, k = 2:
  //  1.99 ms against 2.23; k = 10: 2.46 -> 2.36; 120 rows, k = 16: 1.33 -> 1.12)
  const bool many = NT == 256 && kk <= 32 && kk >= 8 && multisect_many_on;
  if (many) {
    if (wave == 0) tri::multisect_many(dd, e2, na, kk, glo, ghi, lane, lam);
  } else {
    for (int i = wave; i < kk; i += TNW) {
      const int target = na - 1 - i;         // ascending index of the i-th largest eigenvalue
      const double lam_ = tri::multisect(dd, e2, na, target, glo, ghi, lane);
      if (lane == 0) lam[i] = lam_;
    }
  }
  if (all_evals) {                         // the rest of the spectrum (values only), straight to the output
    for (int i = kk + wave; i < na; i += TNW) {
      const int target = na - 1 - i;
      const double lam_ = tri::multisect(dd, e2, na, target, glo, ghi, lane);
      if (lane == 0) evals[i] = lam_ * scale;
    }
    for (int i = na + tid; i < n; i += TNT) evals[i] = 0.0;
  }
  __syncthreads();

  TRI_STAMP(2);
  // ---------------- 3. eigenvectors of T: inverse iteration, one lane per vector ----------------
  // Scratch of the factorisation: [n][ks] arrays.  The register-resident variant keeps them in LDS (the tridiagonalisation
  // buffers are dead by now; the launcher checks the size): the substitutions are chains of dependent loads, and with
  // the scratch in global memory the phase cost 0.27 ms of a 1.27 ms problem in a full batch.
  constexpr bool LSCR = RPW > 0;
  const int ks = LSCR ? kc : kp;
  double *U0, *U1, *U2, *Zg, *Lm, *Ls = nullptr;      // reciprocal pivots; first, second superdiagonal of U (row swaps);
  unsigned char* Lsb = nullptr;                        // rhs / solution; multipliers of L; 1 where rows i, i+1 were swapped
  if constexpr (LSCR) {
    double* b = lam + 72 + n;
    U0 = b;
    U1 = b + (size_t)n * ks;
    U2 = b + (size_t)2 * n * ks;
    Zg = b + (size_t)3 * n * ks;
    Lm = b + (size_t)4 * n * ks;
    Lsb = reinterpret_cast<unsigned char*>(b + (size_t)5 * n * ks);
  } else {
    U0 = scr;
    U1 = scr + (size_t)n * kp;
    U2 = scr + (size_t)2 * n * kp;
    Zg = scr + (size_t)3 * n * kp;
    Lm = scr + (size_t)4 * n * kp;
    Ls = scr + (size_t)5 * n * kp;
  }
  auto swapped = [&](int i, int c) -> bool {
    if constexpr (LSCR) return Lsb[i * ks + c] != 0;
    else return Ls[(size_t)i * ks + c] != 0.0;
  };
  // vectors in chunks of kc (= all of them unless the LDS scratch is short: the launcher sizes the chunk); the solutions
  // of a chunk move to registers (one wave per vector) before the next chunk reuses the scratch
  constexpr int VPW = 4;                    // vectors per wave: c = wave + TNW v
  double z[VPW][RPL];
#pragma unroll
  for (int v = 0; v < VPW; ++v)
#pragma unroll
    for (int rr = 0; rr < RPL; ++rr) z[v][rr] = 0.0;
  for (int c0 = 0; c0 < kk; c0 += kc) {
  if (tid < kc && c0 + tid < kk) {
    const int c = c0 + tid, cs = tid;      // vector, scratch column
    // distinct shifts for (nearly) equal eigenvalues; the offsets are far below the eigenvalue accuracy that matters
    const double lc = lam[c] - (double)(c + 1) * 4.0 * TEPS;
    const double ptiny = 1e-3 * TEPS;      // pivot floor (scaled matrix has max-norm 1)
    // factorisation P L U = T - lc I with partial pivoting (done once), first right-hand side on the fly
    double p = dd[0] - lc, q = (na > 1) ? ee[0] : 0.0, r = 0.0;
    double yc = hash_unit(0u, (unsigned)c);
    for (int i = 0; i + 1 < na; ++i) {
      const double sub = ee[i], nd = dd[i + 1] - lc, nu = (i + 2 < na) ? ee[i + 1] : 0.0;
      const double yn = hash_unit((unsigned)(i + 1), (unsigned)c);
      double inv, u1, u2, yi, m;
      bool sw;
      if (fabs(sub) > fabs(p) && fabs(sub) >= ptiny) {         // swap rows i and i+1
        inv = fast_rcp(sub);
        u1 = nd; u2 = nu;
        m = p * inv;
        sw = true;
        yi = yn;
        yc = yc - m * yn;
        p = q - m * nd;
        q = r - m * nu;
        r = 0.0;
      } else {
        if (fabs(p) < ptiny) p = (p < 0.0) ? -ptiny : ptiny;
        inv = fast_rcp(p);
        u1 = q; u2 = r;
        m = sub * inv;
        sw = false;
        yi = yc;
        yc = yn - m * yc;
        p = nd - m * q;
        q = nu - m * r;
        r = 0.0;
      }
      U0[(size_t)i * ks + cs] = inv;
      U1[(size_t)i * ks + cs] = u1;
      U2[(size_t)i * ks + cs] = u2;
      Lm[(size_t)i * ks + cs] = m;
      if constexpr (LSCR) Lsb[i * ks + cs] = sw ? 1 : 0;
      else Ls[(size_t)i * ks + cs] = sw ? 1.0 : 0.0;
      Zg[(size_t)i * ks + cs] = yi;
    }
    if (fabs(p) < ptiny) p = (p < 0.0) ? -ptiny : ptiny;
    const double invlast = fast_rcp(p);
    double rs = 1.0;
    for (int it = 0; it < 2; ++it) {
      if (it > 0 && na > 1) {
        // forward substitution of the previous solution (scaled to unit norm) through P L; the operands of row i+1
        // are fetched while row i is computed
        yc = Zg[cs] * rs;
        double ynr = Zg[(size_t)ks + cs], m = Lm[cs];
        bool sw = swapped(0, cs);
        for (int i = 0; i + 1 < na; ++i) {
          const double yn = ynr * rs, mi = m;
          const bool swi = sw;
          if (i + 2 < na) {
            ynr = Zg[(size_t)(i + 2) * ks + cs];
            m = Lm[(size_t)(i + 1) * ks + cs];
            sw = swapped(i + 1, cs);
          }
          const double yi = swi ? yn : yc;
          yc = swi ? (yc - mi * yn) : (yn - mi * yc);
          Zg[(size_t)i * ks + cs] = yi;
        }
      } else if (it > 0) {
        yc = Zg[cs] * rs;
      }
      // back substitution (same prefetch)
      double x1 = yc * invlast, x2 = 0.0;
      Zg[(size_t)(na - 1) * ks + cs] = x1;
      double acc = x1 * x1;
      double zn = 0.0, u1n = 0.0, u2n = 0.0, u0n = 0.0;
      if (na > 1) {
        zn = Zg[(size_t)(na - 2) * ks + cs];
        u1n = U1[(size_t)(na - 2) * ks + cs];
        u2n = U2[(size_t)(na - 2) * ks + cs];
        u0n = U0[(size_t)(na - 2) * ks + cs];
      }
      for (int i = na - 2; i >= 0; --i) {
        const double z = zn, u1 = u1n, u2 = u2n, u0 = u0n;
        if (i > 0) {
          zn = Zg[(size_t)(i - 1) * ks + cs];
          u1n = U1[(size_t)(i - 1) * ks + cs];
          u2n = U2[(size_t)(i - 1) * ks + cs];
          u0n = U0[(size_t)(i - 1) * ks + cs];
        }
        const double x = (z - u1 * x1 - u2 * x2) * u0;
        Zg[(size_t)i * ks + cs] = x;
        acc += x * x;
        x2 = x1;
        x1 = x;
      }
      rs = acc > 0.0 ? 1.0 / sqrt(acc) : 1.0;
    }
    // the last solution is left un-normalised: Gram-Schmidt normalises
  }
  __syncthreads();

#pragma unroll
  for (int v = 0; v < VPW; ++v) {
    const int c = wave + TNW * v;
    if (c >= c0 && c < c0 + kc && c < kk) {
#pragma unroll
      for (int rr = 0; rr < RPL; ++rr) {
        const int i = lane + 64 * rr;
        z[v][rr] = (i < na) ? Zg[(size_t)i * ks + (c - c0)] : 0.0;
      }
    }
  }
  __syncthreads();
  }   // chunks of vectors

  TRI_STAMP(3);
  // ---------------- 3b. modified Gram-Schmidt, one wave per vector (registers), pivot vector through LDS -------------
  double* qv = pcur;                        // pivot vector [n]
  for (int c = 0; c < kk; ++c) {
    const int ow = c % TNW, ov = c / TNW;
    if (wave == ow) {
#pragma unroll
      for (int v = 0; v < VPW; ++v)
        if (v == ov) {
          double s = 0.0;
#pragma unroll
          for (int rr = 0; rr < RPL; ++rr) s += z[v][rr] * z[v][rr];
          s = wave_sum(s);
          const double inv = s > 0.0 ? 1.0 / sqrt(s) : 0.0;
#pragma unroll
          for (int rr = 0; rr < RPL; ++rr) {
            z[v][rr] *= inv;
            const int i = lane + 64 * rr;
            if (i < n) qv[i] = z[v][rr];
          }
        }
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < VPW; ++v) {
      const int c2 = wave + TNW * v;
      if (c2 > c && c2 < kk) {
        double s = 0.0;
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) {
          const int i = lane + 64 * rr;
          s += (i < n) ? z[v][rr] * qv[i] : 0.0;
        }
        s = wave_sum(s);
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) {
          const int i = lane + 64 * rr;
          if (i < n) z[v][rr] -= s * qv[i];
        }
      }
    }
    __syncthreads();
  }

  TRI_STAMP(4);
  // ---------------- 4. back-transformation: reflectors staged through LDS in blocks, one wave per vector ------------
  {
    constexpr int RB = 6;                             // reflectors per block: RB * n doubles of LDS (vcur..ee reused;
                                                      // tau, which is still needed, starts at 6n)
    double* stage = sm;                               // [RB][n]; the tridiagonalisation vectors are dead by now
    const int nv = (kk - wave + TNW - 1) / TNW;      // vectors of this wave (may be <= 0)
    for (int jb = na - 3; jb >= 0; jb -= RB) {
      __syncthreads();
      for (int e = tid; e < RB * n; e += TNT) {
        const int q = e / n, i = e - q * n, j = jb - q;
        stage[e] = (j >= 0 && i > j && i < na) ? A[(size_t)j * n + i] : 0.0;
      }
      __syncthreads();
      for (int q = 0; q < RB; ++q) {
        const int j = jb - q;
        if (j < 0) break;
        const double beta = tau[j];
        double vj[RPL];
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) {
          const int i = lane + 64 * rr;
          vj[rr] = (i < n) ? stage[q * n + i] : 0.0;
        }
#pragma unroll
        for (int v = 0; v < VPW; ++v) {
          if (v < nv) {
            double sdot = 0.0;
#pragma unroll
            for (int rr = 0; rr < RPL; ++rr) sdot += vj[rr] * z[v][rr];
            sdot = beta * wave_sum(sdot);
#pragma unroll
            for (int rr = 0; rr < RPL; ++rr) z[v][rr] -= sdot * vj[rr];
          }
        }
      }
    }
  }

  // ---------------- output: sign convention of eigh.hip, zero padding ----------------
#pragma unroll
  for (int v = 0; v < VPW; ++v) {
    const int c = wave + TNW * v;
    if (c < k) {
      double best = -1.0, bval = 0.0;
      int bidx = 0x7fffffff;
#pragma unroll
      for (int rr = 0; rr < RPL; ++rr) {
        const int i = lane + 64 * rr;
        const double a = fabs(z[v][rr]);
        if (i < na && a > best) {
          best = a;
          bval = z[v][rr];
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
      for (int rr = 0; rr < RPL; ++rr) {
        const int i = lane + 64 * rr;
        if (i < n) evecs[(size_t)c * n + i] = (i < na) ? z[v][rr] * sg : 0.0;
      }
      if (lane == 0) evals[c] = (c < kk) ? lam[c] * scale : 0.0;
    }
  }
  TRI_STAMP(5);
  if (guard != nullptr && fail != nullptr && prob == 0 && tid == 0) {
    atomicSub(fail + 1, (int)__hip_atomic_load(gcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    atomicAdd(fail + 3, 1);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-workgroup variant for a single (or a few) larger problems: a lone CU streams the trailing matrix from L2 at
// ~50 GB/s, which makes the tridiagonalisation of one 400 x 400 matrix take 7 ms.  Here W workgroups hold the matrix
// in LDS (row r lives in workgroup r mod W) and exchange, per Householder step, only two vectors through global memory
// behind ONE counter barrier:
//     pass s   : own rows r > s:  row <- row - v'_r w' - w'_r v'  (pending update of step s-1),
//                p[r] = beta_s row . v_s,   col[r] = row[s+1]            -> global, write-through
//     barrier ; every workgroup gathers p and col (= row s+1 by symmetry), forms w_s = p - K v_s, applies the step's
//     own update to row s+1 and derives v_{s+1}, beta_{s+1} redundantly -- no broadcast step.
// Afterwards every workgroup computes the eigenvalues / inverse iterations / back-transformations of ITS vectors
// (vector c belongs to workgroup c mod W, all scratch in LDS), one more barrier, and workgroup 0 orthonormalises.
template <int RPL>
__global__ __launch_bounds__(TNT) void tri_multi_kernel(double* __restrict__ Aall, int n, int k, int RW, int VW, int rows_d, int all_evals,
                                                        double* __restrict__ evals_all, double* __restrict__ evecs_all,
                                                        double* __restrict__ gbuf_all, unsigned* __restrict__ bars, int one_xcd,
                                                        int* __restrict__ fail, const double* __restrict__ det = nullptr,
                                                        const unsigned* __restrict__ dead = nullptr, int drop_wg = -1) {
  extern __shared__ double sm[];
  const int prob = blockIdx.y;
  // dead: the wave-resident reduction that produced `det` timed out (bar[4] of tri_wave_kernel): det and the reflectors in A are
  // void -- leave (the recovery launch that follows solves the problem again); read before any barrier, the same for everybody
  if (dead != nullptr && __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  // one_xcd = 1 + base: the grid is 8 x wider and only the ids that land on XCD (base + problem) % 8 stay (ids go round-robin
  // over the XCDs):
  // the W workgroups of a problem then share one L2 and exchange through it (wave_util.h, checked below)
  if (one_xcd && (int)(blockIdx.x & 7) != ((one_xcd - 1 + prob) & 7)) return;
  const int W = one_xcd ? gridDim.x >> 3 : gridDim.x, wg = one_xcd ? blockIdx.x >> 3 : blockIdx.x;
  if (wg == drop_wg) return;                 // (test hook, option eigh_multi_drop: a participant that never arrives -- the others time out)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* rows = sm;                         // [RW][n]  row lr <-> global row lr*W + wg ; reused after phase 1
  double* vbuf0 = rows + rows_d;             // Householder vector, double buffered (rows_d >= RW*n, 6*n*VW)
  double* vbuf1 = vbuf0 + n;
  double* vprev = vbuf1 + n;
  double* wprev = vprev + n;
  double* pfull = wprev + n;
  double* cfull = pfull + n;
  double* dd = cfull + n;
  double* ee = dd + n;
  double* tau = ee + n;
  double* e2 = tau + n;
  double* lam = e2 + n;                      // [64]
  double* A = Aall + (size_t)prob * n * n;
  double* evals = evals_all + (size_t)prob * n;
  double* evecs = evecs_all + (size_t)prob * n * n;
  double* gb = gbuf_all + (size_t)prob * 5 * n;
  double* Pb = gb;                           // [2][n]
  double* Cb = gb + 2 * n;                   // [2][n]
  double* Db = gb + 4 * n;                   // [n] diagonal, last step only
  unsigned* bar = bars + (size_t)prob * TRI_BAR_WORDS;       // [0] counter, [8 ..] flags, [72 ..] XCC ids
  unsigned* xflags = bar + 8;
  unsigned* xids = bar + 72;
  unsigned bar_target = 0, xepoch = 0;
  const int na = n;
  const int kk = k < na ? k : na;

  // det != nullptr: the tridiagonalisation was done by tri_wave_reduce (eigh_wave.hip): d, e, tau come from det[3][n], the
  // reflectors are in A; this launch only runs stages 2-5 (agent-scope exchange: one barrier, nothing to gain from the XCD path)
  const bool pre = det != nullptr;
  // own rows into LDS; row 0 (input data, no synchronisation needed) gives v_0 in every workgroup
  for (int e = tid; e < (pre ? 0 : RW * n); e += TNT) {
    const int lr = e / n, c = e - lr * n, r = lr * W + wg;
    rows[e] = (r < n) ? A[(size_t)r * n + c] : 0.0;
  }
  for (int c = tid; c < n; c += TNT) {
    cfull[c] = pre ? 0.0 : A[c];
    vprev[c] = 0.0;
    wprev[c] = 0.0;
    pfull[c] = 0.0;
    if (pre) {
      dd[c] = det[c];
      ee[c] = det[n + c];
      tau[c] = det[2 * n + c];
    }
  }
  // every workgroup has read row 0 (and its own rows) before any reflector is written over the input matrix
  if (!pre) {
    if (one_xcd && tid == 0) __hip_atomic_store(xids + wg, 1u + xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bar_target += W;
    grid_barrier(bar, bar_target, W, fail);
  }
  bool fast = one_xcd != 0 && !pre;          // all workgroups really on one XCD?  (uniform: everybody reads the same ids)
  if (fast) {
    const unsigned mine = 1u + xcc_id();
    for (int j = 0; j < W; ++j) fast = fast && __hip_atomic_load(xids + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine;
  }
  auto put = [&](double* p, double v) {
    if (fast) st_xcd(p, v);
    else st_shared(p, v);
  };
  auto sync_all = [&]() {
    if (fast) {
      xepoch += 1;
      xcd_barrier(xflags, xepoch, W, wg, fail);
    } else {
      bar_target += W;
      grid_barrier(bar, bar_target, W, fail);
    }
  };
  double* vcur = vbuf0;
  double* vnext = vbuf1;
  // derive (v_{s+1}, beta, alpha, diagonal) from x = updated row s+1 held in cfull[c], c >= s+1 ; every wave
  // computes the norm redundantly, threads write disjoint elements
  auto form_reflector = [&](int s1, double* vout) {      // s1 = index of the row that becomes the Householder column
    double nrm2 = 0.0;
    for (int c = s1 + 1 + lane; c < na; c += 64) {
      const double x = cfull[c];
      nrm2 += x * x;
    }
    nrm2 = wave_sum(nrm2);
    const double x0 = cfull[s1 + 1];
    const double nrm = sqrt(nrm2);
    const double alpha = (x0 >= 0.0) ? -nrm : nrm;
    const double v0 = x0 - alpha;
    double rest = nrm2 - x0 * x0;
    if (rest < 0.0) rest = 0.0;
    const double vv = rest + v0 * v0;
    const double beta = (nrm2 > 0.0 && vv > 0.0) ? 2.0 / vv : 0.0;
    for (int c = s1 + 1 + tid; c < na; c += TNT) vout[c] = (c == s1 + 1) ? v0 : cfull[c];
    if (tid == 0) {
      dd[s1] = cfull[s1];
      ee[s1] = (nrm2 > 0.0) ? alpha : 0.0;
      tau[s1] = beta;
    }
  };
  if (!pre) form_reflector(0, vcur);
  __syncthreads();

  // ---------------- 1. tridiagonalisation ----------------
#ifdef VIPMI_TRI_PROFILE      // stage stamps of workgroup 0 (s_memtime; tools/multi_profile.py): evals[n-16 ..] at the end
  unsigned long long pst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PSTAMP(i) pst[i] = __builtin_amdgcn_s_memtime()
#else
#define PSTAMP(i)
#endif
#ifdef VIPMI_TRI_PROFILE      // s_memtime segments of wave 0 of workgroup 0 (tools/multi_profile.py)
  unsigned long long seg_t[6] = {0, 0, 0, 0, 0, 0}, seg_c = __builtin_amdgcn_s_memtime();
#define MSEG(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg_t[i] += t_ - seg_c; seg_c = t_; } while (0)
#else
#define MSEG(i)
#endif
  for (int s = 0; s + 2 < (pre ? 0 : na); ++s) {
    const int par = s & 1;
    const double beta = tau[s];
    // the owner of row s keeps the reflector for the back-transformation
    if (s % W == wg)
      for (int c = s + 1 + tid; c < na; c += TNT) put(&A[(size_t)s * n + c], vcur[c]);
    const int lr0 = (s + 1 - wg + W - 1) / W;            // first local row with r >= s+1
    for (int lr = (lr0 > 0 ? lr0 : 0) + wave; lr < RW; lr += TNW) {
      const int r = lr * W + wg;
      if (r >= na) break;
      double* row = rows + (size_t)lr * n;
      const double vr = vprev[r], wr = wprev[r];
      double acc = 0.0, cval = 0.0, dval = 0.0;
#pragma unroll
      for (int ch = 0; ch < RPL; ++ch) {
        const int c = s + 1 + lane + 64 * ch;
        if (c < na) {
          const double t = row[c] - vr * wprev[c] - wr * vprev[c];
          row[c] = t;
          acc += t * vcur[c];
          if (ch == 0 && lane == 0) cval = t;
          if (c == r) dval = t;
        }
      }
      acc = wave_sum(acc);
      if (lane == 0) {
        put(&Pb[par * n + r], beta * acc);
        put(&Cb[par * n + r], cval);
      }
      if (s + 3 == na) {
        const int dl = r - s - 1;
        if (lane == (dl & 63)) put(&Db[r], dval);
      }
    }
    MSEG(0);
    sync_all();
    MSEG(1);
    for (int c = s + 1 + tid; c < na; c += TNT) {
      pfull[c] = ld_shared(&Pb[par * n + c]);
      cfull[c] = ld_shared(&Cb[par * n + c]);
    }
    __syncthreads();
    MSEG(2);
    double kd = 0.0;
    for (int r = s + 1 + lane; r < na; r += 64) kd += vcur[r] * pfull[r];
    const double K = 0.5 * beta * wave_sum(kd);
    const double vs1 = vcur[s + 1], ws1 = pfull[s + 1] - K * vs1;
    for (int r = s + 1 + tid; r < na; r += TNT) {
      const double v = vcur[r];
      const double w = pfull[r] - K * v;
      wprev[r] = w;
      vprev[r] = v;
      cfull[r] = cfull[r] - vs1 * w - ws1 * v;          // row s+1 with this step's update
    }
    __syncthreads();
    MSEG(3);
    if (s + 3 < na) {
      form_reflector(s + 1, vnext);
      double* t = vcur;
      vcur = vnext;
      vnext = t;
    }
    __syncthreads();
    MSEG(4);
  }
#ifdef VIPMI_TRI_PROFILE
  if (wg == 0 && tid == 0 && prob == 0)
    for (int i = 0; i < 5; ++i) evals[n - 8 + i] = (double)seg_t[i];
#endif
  if (tid == 0 && !pre) {
    // trailing 2 x 2 block: cfull holds row na-2 (fully updated) ; the last diagonal entry came through Db
    const int a = na - 2, b = na - 1;
    dd[a] = cfull[a];
    ee[a] = cfull[b];
    dd[b] = ld_shared(&Db[b]) - 2.0 * vprev[b] * wprev[b];
    ee[b] = 0.0;
  }
  __syncthreads();

  // ---------------- 2. eigenvalues of the vectors of this workgroup ----------------
  PSTAMP(0);
  double scale = 0.0, glo = 0.0, ghi = 0.0;
  {
    double mx = 0.0;
    for (int i = lane; i < na; i += 64) mx = fmax(mx, fmax(fabs(dd[i]), fabs(ee[i])));
    scale = wave_max(mx);
  }
  const double iscale = scale > 0.0 ? 1.0 / scale : 0.0;
  __syncthreads();
  for (int i = tid; i < na; i += TNT) {
    const double e = ee[i] * iscale;
    dd[i] *= iscale;
    ee[i] = e;
    e2[i] = e * e;
  }
  __syncthreads();
  {
    double lo = 1e300, hi = -1e300;
    for (int i = lane; i < na; i += 64) {
      const double rad = (i > 0 ? fabs(ee[i - 1]) : 0.0) + (i + 1 < na ? fabs(ee[i]) : 0.0);
      lo = fmin(lo, dd[i] - rad);
      hi = fmax(hi, dd[i] + rad);
    }
    glo = -wave_max(-lo);
    ghi = wave_max(hi);
    const double margin = 4.0 * TEPS * (double)na + 1e-290;
    glo -= margin;
    ghi += margin;
  }
  const int nmine = (kk - wg + W - 1) / W;               // vectors c = wg + W j, j < nmine (may be <= 0)
  for (int j = wave; j < nmine; j += TNW) {
    const int c = wg + W * j;
    const int target = na - 1 - c;
    const double lam_ = tri::multisect(dd, e2, na, target, glo, ghi, lane);
    if (lane == 0) lam[j] = lam_;
  }
  if (all_evals) {                         // the rest of the spectrum (values only), spread over all waves
    for (int i = kk + wg * TNW + wave; i < na; i += W * TNW) {
      const int target = na - 1 - i;
      const double lam_ = tri::multisect(dd, e2, na, target, glo, ghi, lane);
      if (lane == 0) evals[i] = lam_ * scale;
    }
  }
  __syncthreads();

  // ---------------- 3. inverse iteration, vectors of this workgroup, everything in LDS (rows region) ----------------
  // VW = vectors per workgroup (ceil(k / W)) = row stride of the per-vector arrays
  PSTAMP(1);
  double* U0 = rows;                                      // [n][VW]
  double* U1 = U0 + (size_t)n * VW;
  double* U2 = U1 + (size_t)n * VW;
  double* Lm = U2 + (size_t)n * VW;
  double* Ls = Lm + (size_t)n * VW;
  double* Zl = Ls + (size_t)n * VW;
  if (tid < nmine) {
    const int j = tid, c = wg + W * j;
    const double lc = lam[j] - (double)(c + 1) * 4.0 * TEPS;
    const double ptiny = 1e-3 * TEPS;
    double p = dd[0] - lc, q = (na > 1) ? ee[0] : 0.0, r = 0.0;
    double yc = hash_unit(0u, (unsigned)c);
    for (int i = 0; i + 1 < na; ++i) {
      const double sub = ee[i], nd = dd[i + 1] - lc, nu = (i + 2 < na) ? ee[i + 1] : 0.0;
      const double yn = hash_unit((unsigned)(i + 1), (unsigned)c);
      double inv, u1, u2, yi, m, sw;
      if (fabs(sub) > fabs(p) && fabs(sub) >= ptiny) {
        inv = fast_rcp(sub);
        u1 = nd; u2 = nu;
        m = p * inv;
        sw = 1.0;
        yi = yn;
        yc = yc - m * yn;
        p = q - m * nd;
        q = r - m * nu;
        r = 0.0;
      } else {
        if (fabs(p) < ptiny) p = (p < 0.0) ? -ptiny : ptiny;
        inv = fast_rcp(p);
        u1 = q; u2 = r;
        m = sub * inv;
        sw = 0.0;
        yi = yc;
        yc = yn - m * yc;
        p = nd - m * q;
        q = nu - m * r;
        r = 0.0;
      }
      U0[i * VW + j] = inv;
      U1[i * VW + j] = u1;
      U2[i * VW + j] = u2;
      Lm[i * VW + j] = m;
      Ls[i * VW + j] = sw;
      Zl[i * VW + j] = yi;
    }
    if (fabs(p) < ptiny) p = (p < 0.0) ? -ptiny : ptiny;
    const double invlast = fast_rcp(p);
    double rs = 1.0;
    for (int it = 0; it < 2; ++it) {
      if (it > 0) {
        yc = Zl[j] * rs;
        for (int i = 0; i + 1 < na; ++i) {
          const double yn = Zl[(i + 1) * VW + j] * rs;
          const double m = Lm[i * VW + j];
          const bool sw = Ls[i * VW + j] != 0.0;
          const double yi = sw ? yn : yc;
          yc = sw ? (yc - m * yn) : (yn - m * yc);
          Zl[i * VW + j] = yi;
        }
      }
      double x1 = yc * invlast, x2 = 0.0;
      Zl[(na - 1) * VW + j] = x1;
      double acc = x1 * x1;
      for (int i = na - 2; i >= 0; --i) {
        const double x = (Zl[i * VW + j] - U1[i * VW + j] * x1 - U2[i * VW + j] * x2) * U0[i * VW + j];
        Zl[i * VW + j] = x;
        acc += x * x;
        x2 = x1;
        x1 = x;
      }
      rs = acc > 0.0 ? 1.0 / sqrt(acc) : 1.0;
    }
    lam[32 + j] = rs;                                     // normalisation of the final solution
  }
  __syncthreads();

  // ---------------- 4. back-transformation of this workgroup's vectors: wave j <-> vector j ----------------
  PSTAMP(2);
  double z[RPL];
  const bool have = wave < nmine;
  {
    const double rs = have ? lam[32 + wave] : 0.0;
#pragma unroll
    for (int rr = 0; rr < RPL; ++rr) {
      const int i = lane + 64 * rr;
      z[rr] = (have && i < na) ? Zl[i * VW + wave] * rs : 0.0;
    }
  }
  {
    constexpr int RB = 4;                                 // reflectors per staged block
    double* stage = vbuf0;                                // [RB][n] over vbuf0, vbuf1, vprev, wprev (dead)
    for (int jb = na - 3; jb >= 0; jb -= RB) {
      __syncthreads();
      for (int e = tid; e < RB * n; e += TNT) {
        const int q = e / n, i = e - q * n, j = jb - q;
        stage[e] = (j >= 0 && i > j && i < na) ? ld_shared(&A[(size_t)j * n + i]) : 0.0;
      }
      __syncthreads();
      if (have) {
        for (int q = 0; q < RB; ++q) {
          const int j = jb - q;
          if (j < 0) break;
          const double beta = tau[j];
          double vj[RPL], sdot = 0.0;
#pragma unroll
          for (int rr = 0; rr < RPL; ++rr) {
            const int i = lane + 64 * rr;
            vj[rr] = (i < n) ? stage[q * n + i] : 0.0;
            sdot += vj[rr] * z[rr];
          }
          sdot = beta * wave_sum(sdot);
#pragma unroll
          for (int rr = 0; rr < RPL; ++rr) z[rr] -= sdot * vj[rr];
        }
      }
    }
  }
  if (have) {
    const int c = wg + W * wave;
#pragma unroll
    for (int rr = 0; rr < RPL; ++rr) {
      const int i = lane + 64 * rr;
      if (i < n) put(&evecs[(size_t)c * n + i], z[rr]);
    }
    if (lane == 0) put(&evals[c], lam[wave] * scale);
  }
  PSTAMP(3);
  sync_all();
  PSTAMP(4);
  if (wg != 0) return;

  // ---------------- 5. workgroup 0: modified Gram-Schmidt over the k vectors, sign convention, output ----------------
  constexpr int VPW = 4;
  double zz[VPW][RPL];
#pragma unroll
  for (int v = 0; v < VPW; ++v) {
    const int c = wave + TNW * v;
#pragma unroll
    for (int rr = 0; rr < RPL; ++rr) {
      const int i = lane + 64 * rr;
      zz[v][rr] = (c < kk && i < na) ? ld_shared(&evecs[(size_t)c * n + i]) : 0.0;
    }
  }
  double* qv = pfull;
  for (int c = 0; c < kk; ++c) {
    const int ow = c % TNW, ov = c / TNW;
    if (wave == ow) {
#pragma unroll
      for (int v = 0; v < VPW; ++v)
        if (v == ov) {
          double sq = 0.0;
#pragma unroll
          for (int rr = 0; rr < RPL; ++rr) sq += zz[v][rr] * zz[v][rr];
          sq = wave_sum(sq);
          const double inv = sq > 0.0 ? 1.0 / sqrt(sq) : 0.0;
#pragma unroll
          for (int rr = 0; rr < RPL; ++rr) {
            zz[v][rr] *= inv;
            const int i = lane + 64 * rr;
            if (i < n) qv[i] = zz[v][rr];
          }
        }
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < VPW; ++v) {
      const int c2 = wave + TNW * v;
      if (c2 > c && c2 < kk) {
        double sq = 0.0;
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) {
          const int i = lane + 64 * rr;
          sq += (i < n) ? zz[v][rr] * qv[i] : 0.0;
        }
        sq = wave_sum(sq);
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) {
          const int i = lane + 64 * rr;
          if (i < n) zz[v][rr] -= sq * qv[i];
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int v = 0; v < VPW; ++v) {
    const int c = wave + TNW * v;
    if (c < k) {
      double best = -1.0, bval = 0.0;
      int bidx = 0x7fffffff;
#pragma unroll
      for (int rr = 0; rr < RPL; ++rr) {
        const int i = lane + 64 * rr;
        const double a = fabs(zz[v][rr]);
        if (i < na && a > best) {
          best = a;
          bval = zz[v][rr];
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
      for (int rr = 0; rr < RPL; ++rr) {
        const int i = lane + 64 * rr;
        if (i < n) evecs[(size_t)c * n + i] = (i < na) ? zz[v][rr] * sg : 0.0;
      }
      if (lane == 0 && c >= kk) evals[c] = 0.0;
    }
  }
#ifdef VIPMI_TRI_PROFILE
  PSTAMP(5);
  if (tid == 0 && prob == 0)
    for (int i = 0; i < 6; ++i) evals[n - 16 + i] = (double)(pst[i] - pst[0]);
#endif
}

// The recovery launch of a cooperating solve (see tri_eig_kernel, `guard`): one workgroup per problem on the LDS-resident
// single-workgroup solver, from the copy `Abak` of the input (destroyed).  A few microseconds in the stream when the solve went through.
template <int RPL>
int launch_recovery(vipmi_ctx* ctx, double* Abak, int64_t batch, int n, int k, double* evals, double* evecs, int all_evals,
                    const unsigned* guard, const unsigned* gcount, int* fail) {
  static_assert(RPL == 2 || RPL == 4 || RPL == 8, "launch_recovery: up to 512 rows");
  const int kp = (int)cdiv(k, 16) * 16;
  double* scratch = nullptr;
  VIPMI_TRY(ws(ctx, "eigh_fb_scratch", (size_t)batch * 6 * n * kp, &scratch));
  // 512 threads up to 32 vectors (the Gram-Schmidt stage holds four per wave): the 1024-thread instance of the 512-row layout
  // spills (96 bytes of scratch per lane), and a dispatch that needs scratch costs ~20 us even when its workgroup leaves in the
  // first instruction -- on the critical path of every lone synchronous solve
  const int nt = k <= 32 ? 512 : 1024;
  const size_t lds = ((size_t)(9 + nt / 64) * n + 64 + 8) * sizeof(double);
  if (nt == 512) {
    auto kern = tri_eig_kernel<RPL, 512>;
    VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(512), lds, ctx->stream, Abak, n, k, (const int32_t*)nullptr, evals, evecs,
                       scratch, kp, all_evals, k < n ? k : n, 0, (double*)nullptr, guard, gcount, fail, TriGather());
  } else {
    auto kern = tri_eig_kernel<RPL, 1024>;
    VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(1024), lds, ctx->stream, Abak, n, k, (const int32_t*)nullptr, evals, evecs,
                       scratch, kp, all_evals, k < n ? k : n, 0, (double*)nullptr, guard, gcount, fail, TriGather());
  }
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

template <int RPL>
int launch_tri_multi(vipmi_ctx* ctx, double* A, int64_t batch, int n, int k, double* evals, double* evecs,
                     int all_evals) {
  // a copy of the input for the recovery launch (the cooperating kernels overwrite A with their reflectors): n = 400: 1.3 MB
  constexpr bool RECOVER = (RPL == 2 || RPL == 4 || RPL == 8);
  double* Abak = nullptr;
  if (RECOVER && ctx->opt("eigh_recover", 1) != 0) {
    VIPMI_TRY(ws(ctx, "eigh_fb_A", (size_t)batch * n * n, &Abak));
    VIPMI_CHECK_HIP(hipMemcpyAsync(Abak, A, sizeof(double) * (size_t)batch * n * n, hipMemcpyDeviceToDevice, ctx->stream));
  }
  int W = n <= 256 ? 8 : (n <= 448 ? 16 : 32);
  // a lone synchronous call owns the chip: with the workgroups of a problem on ONE XCD (below) twice as many of them halve the
  // row pass at no extra exchange cost (n = 400: 16 / 24 / 32 workgroups 1.74 / 1.66 / 1.63 ms, round 3); the pipelined mode keeps
  // 16 -- its eigensolver runs beside the other call's shears and every CU it takes is one they lose
  // (option eigh_wave_async, experiment of round 5: the wave-resident path also in the pipelined mode, where its 64 waves must find
  //  their SIMDs between the other call's persistent shear workgroups: 1 = on one XCD, 2 = spread over the chip)
  const int64_t wave_async = ctx->opt("eigh_check", 1) == 0 ? ctx->opt("eigh_wave_async", 0) : 0;
  const bool lone = batch == 1 && (ctx->opt("eigh_check", 1) != 0 || wave_async != 0) && ctx->opt("eigh_one_xcd", -1) != 0 && ctx->num_cu % 8 == 0;
  if (lone && n > 320 && n <= 448 && ctx->num_cu / 8 >= 32 && wave_async == 0) W = 32;
  if (ctx->opt("eigh_w", 0) >= 2 && ctx->opt("eigh_w", 0) <= 64) W = (int)ctx->opt("eigh_w", 0);       // (experiments)
  const int RW = (int)cdiv(n, W);
  double* gbuf = nullptr;
  unsigned* bars = nullptr;
  VIPMI_TRY(ws(ctx, "eigh_tri_gbuf", (size_t)batch * 5 * n, &gbuf));
  VIPMI_TRY(ws(ctx, "eigh_tri_bars", (size_t)batch * TRI_BAR_WORDS, &bars));
  // (the wave-resident path below needs neither this buffer nor the barriers' give-up word unless the whole spectrum is asked for:
  //  two fills less in front of a 1 ms solve)
  const bool wave_only = lone && tri_wave_fits(ctx, n) && ctx->opt("eigh_wave", 1) != 0 && !all_evals && k <= 64 &&
                         W <= 64 && W <= ctx->num_cu / 8 && ctx->opt("eigh_one_xcd", -1) != 0;
  if (!wave_only) VIPMI_CHECK_HIP(hipMemsetAsync(bars, 0, sizeof(unsigned) * batch * TRI_BAR_WORDS, ctx->stream));
  size_t rows_d = (size_t)RW * n;
  const int VW = (int)cdiv(k, W);
  const size_t inv_d = (size_t)6 * n * VW;             // phase-3 scratch aliases the rows region
  if (rows_d < inv_d) rows_d = inv_d;
  const size_t lds = (rows_d + (size_t)10 * n + 64) * sizeof(double);
  VIPMI_REQUIRE(lds <= 160 * 1024, "eigh_topk(multi): LDS budget exceeded (%zu)", lds);
  auto kern = tri_multi_kernel<RPL>;
  VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds));
  // all W workgroups of a problem must be co-resident (counter barrier): one per CU, launch at most num_cu/W problems
  // one_xcd (wave_util.h): problem p on XCD (base + p) % 8, num_cu / 8 CUs each -- the same count.  The base rotates from
  // launch to launch (and starts at a per-process value): concurrent launches from several streams or processes then sit on
  // different XCDs and stay co-resident exactly as the spread layout does (two or four workgroups per XCD each).  Automatic
  // choice: synchronous mode only -- that is where the latency shows and where every pair is verified by its residual
  // afterwards; the pipelined (asynchronous) mode keeps the spread layout, whose CUs the shear kernels of the other calls
  // lose evenly over the XCD task queues.
  static std::atomic<unsigned> xcd_base{(unsigned)getpid() * 2654435761u >> 16};
  const int64_t want_xcd = ctx->opt("eigh_one_xcd", -1);
  const bool xcd_ok = W <= 64 && W <= ctx->num_cu / 8 && ctx->num_cu % 8 == 0;
  const int one_xcd = xcd_ok && (want_xcd > 0 || (want_xcd < 0 && (ctx->opt("eigh_check", 1) != 0 || wave_async != 0))) ? 1 : 0;
  const int64_t per_launch = ctx->num_cu / W > 0 ? ctx->num_cu / W : 1;
  int* fail = nullptr;                                  // barrier time-outs are latched here (vipmi_check_deferred)
  VIPMI_TRY(deferred_fail_words(ctx, &fail, !wave_only));
  auto recover = [&](const unsigned* guard, const unsigned* gcount) -> int {
    if constexpr (RECOVER) {
      if (Abak) return launch_recovery<RPL>(ctx, Abak, batch, n, k, evals, evecs, all_evals, guard, gcount, fail);
    }
    return VIPMI_OK;
  };
  // A lone synchronous problem of 129 .. 448 rows: the tridiagonalisation on 64 single-wave workgroups of one XCD with the
  // matrix in registers (eigh_wave.hip), then stages 2-5 of this kernel as a second launch (option eigh_wave = 0: off)
  if (lone && one_xcd && tri_wave_fits(ctx, n) && ctx->opt("eigh_wave", 1) != 0) {
    double *det = nullptr, *gw = nullptr;
    unsigned* bars2 = nullptr;
    VIPMI_TRY(ws(ctx, "eigh_wave_det", (size_t)3 * n, &det));
    VIPMI_TRY(ws(ctx, "eigh_wave_gbuf", (size_t)4 * 64 * cdiv(n, 64) + 8, &gw));
    VIPMI_TRY(ws(ctx, "eigh_wave_bars", (size_t)TRI_BAR_WORDS, &bars2));
    VIPMI_CHECK_HIP(hipMemsetAsync(bars2, 0, sizeof(unsigned) * TRI_BAR_WORDS, ctx->stream));
    double* gram = nullptr;
    VIPMI_TRY(ws(ctx, "eigh_wave_gram", (size_t)8 * cdiv(n, 4), &gram));
    double* det2 = nullptr;
    VIPMI_TRY(ws(ctx, "eigh_wave_det2", (size_t)3 * n + 8, &det2));
    VIPMI_TRY(tri_wave_reduce(ctx, A, n, det, gw, bars2, wave_async == 2 ? 0 : (int)(1 + (xcd_base.fetch_add(1u) & 7u)), fail, gram, det2));
    // the leading pairs alone: one workgroup per vector (eigh_wave.hip); with the rest of the spectrum: stages 2-5 of this kernel
    if (!all_evals && k <= 64) {
      VIPMI_TRY(tri_wave_vectors(ctx, A, n, k, det, det2, gram, evals, evecs, bars2));
      return recover(bars2 + 4, bars2 + 5);
    }
    // (bars2 + 4: a reduction that timed out leaves det / A void -- stages 2-5 must not run on them)
    hipLaunchKernelGGL(kern, dim3(W, 1), dim3(TNT), lds, ctx->stream, A, n, k, RW, VW, (int)rows_d, all_evals, evals, evecs, gbuf,
                       bars, 0, fail, (const double*)det, (const unsigned*)(bars2 + 4), -1);
    VIPMI_CHECK_HIP(hipGetLastError());
    return recover(bars2 + 4, bars2 + 5);
  }
  // one-XCD layout: ordered against every other cooperating one-XCD launch of the device (common.h: CoopOrder)
  std::unique_ptr<CoopOrder> order;
  if (one_xcd) {
    order.reset(new CoopOrder(ctx));
    VIPMI_TRY(order->status);
  }
  for (int64_t p0 = 0; p0 < batch; p0 += per_launch) {
    const int64_t nb = batch - p0 < per_launch ? batch - p0 : per_launch;
    hipLaunchKernelGGL(kern, dim3(one_xcd ? 8 * W : W, (unsigned)nb), dim3(TNT), lds, ctx->stream, A + (size_t)p0 * n * n, n, k, RW, VW,
                       (int)rows_d, all_evals, evals + (size_t)p0 * n, evecs + (size_t)p0 * n * n, gbuf + (size_t)p0 * 5 * n,
                       bars + (size_t)p0 * TRI_BAR_WORDS, one_xcd ? (int)(1 + ((xcd_base.fetch_add((unsigned)nb) + (unsigned)p0) & 7u)) : 0, fail,
                       (const double*)nullptr, (const unsigned*)nullptr, ctx->opt("eigh_multi_drop", 0) != 0 ? W - 1 : -1);
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  if (order) VIPMI_TRY(order->done());
  // fail[2]: non-zero when a barrier of these launches gave up, and how many time-outs they latched
  return recover(reinterpret_cast<const unsigned*>(fail + 2), reinterpret_cast<const unsigned*>(fail + 2));
}

// LDS of the register-resident variant: the tridiagonalisation buffers (padded vectors, per-wave column sums), later
// overlaid by the inverse-iteration scratch (5 double arrays and one byte array of n x min(k, n))
// vectors per chunk of the inverse iteration: all of them when the scratch fits, else as many as do (at least 8)
int reg_variant_chunk(int n, int k) {
  const int kk_max = k < n ? k : n;
  const size_t lds_fixed = ((size_t)10 * n + 64 + 8 + 72) * sizeof(double);
  const size_t avail = (size_t)160 * 1024 - lds_fixed - 16;
  const int fit = (int)(avail / ((size_t)n * (5 * sizeof(double) + 1)));
  return fit >= kk_max ? kk_max : fit;
}
size_t reg_variant_lds(int n, int k) {
  const size_t lds_fixed = ((size_t)10 * n + 64 + 8 + 72) * sizeof(double);
  const size_t lds_tri = ((size_t)8 * n + 4 * 200 + 16) * sizeof(double);
  const size_t lds_inv = (size_t)n * reg_variant_chunk(n, k) * (5 * sizeof(double) + 1) + 16;
  return lds_fixed + (lds_tri > lds_inv ? lds_tri : lds_inv);
}
bool reg_variant_fits(int n, int k) { return k <= 32 && cdiv(n, 8) <= 25 && reg_variant_chunk(n, k) >= (k < 8 ? k : 8); }

template <int RPL>
int launch_tri(vipmi_ctx* ctx, double* A, int64_t batch, int n, int k, const int32_t* nact, double* evals,
               double* evecs, int all_evals, const TriGather& gat = TriGather()) {
  const int kp = (int)cdiv(k, 16) * 16;
  double* scratch = nullptr;
  VIPMI_TRY(ws(ctx, "eigh_tri_scratch", (size_t)batch * 6 * n * kp, &scratch));
  // workgroup size: several problems per CU once there are more problems than CUs (option eigh_nt overrides)
  int nt = (int)ctx->opt("eigh_nt", 0);
  if (nt != 256 && nt != 512 && nt != 1024) nt = batch > ctx->num_cu ? 512 : 1024;   // measured: 256 never wins
  while (nt < 1024 && k > nt / 16) nt *= 2;
  // register-resident tridiagonalisation (512 threads: 8 waves x RPW rows x RPL x 64 columns) whenever the matrix fits
  const int rpw_need = (int)cdiv(n, 8);
  const size_t lds_r = reg_variant_lds(n, k);
  const bool reg = ctx->opt("eigh_reg", 1) != 0 && (RPL == 2 || RPL == 4) && reg_variant_fits(n, k);
  if (reg) {
    auto launch_reg = [&](auto kern) -> int {
      VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds_r));
      // big batches: tridiagonalisation and the rest as two launches (see the kernel); option eigh_split = 0 / 1 forces
      const int64_t split_opt = ctx->opt("eigh_split", -1);
      // (measured, 200 x 200, k = 10: 400 problems 1.47 ms in one launch / 1.57 split, 1600: 4.33 / 3.70, 3200: 8.33 / 6.83)
      const bool split = (split_opt < 0 ? batch >= 4 * ctx->num_cu : split_opt != 0) && k <= 16;
      if (!split) {
        hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(512), lds_r, ctx->stream, A, n, k, nact, evals, evecs, scratch,
                           kp, all_evals, reg_variant_chunk(n, k), 0, (double*)nullptr, (const unsigned*)nullptr, (const unsigned*)nullptr, (int*)nullptr, gat);
        VIPMI_CHECK_HIP(hipGetLastError());
        return VIPMI_OK;
      }
      double* det = nullptr;
      VIPMI_TRY(ws(ctx, "eigh_tri_det", (size_t)batch * 3 * n, &det));
      hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(512), lds_r, ctx->stream, A, n, k, nact, evals, evecs, scratch, kp,
                         all_evals, reg_variant_chunk(n, k), 1, det, (const unsigned*)nullptr, (const unsigned*)nullptr, (int*)nullptr, gat);
      VIPMI_CHECK_HIP(hipGetLastError());
      const size_t lds2 = ((size_t)(9 + 256 / 64) * n + 64 + 8) * sizeof(double);
      auto kern2 = tri_eig_kernel<RPL, 256>;
      VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(kern2), (int)lds2));
      hipLaunchKernelGGL(kern2, dim3((unsigned)batch), dim3(256), lds2, ctx->stream, A, n, k, nact, evals, evecs, scratch, kp,
                         all_evals, k < n ? k : n, ctx->opt("eigh_many", 1) != 0 ? 6 : 2, det, (const unsigned*)nullptr, (const unsigned*)nullptr, (int*)nullptr, TriGather());
      VIPMI_CHECK_HIP(hipGetLastError());
      return VIPMI_OK;
    };
    if constexpr (RPL == 2) {
      if (rpw_need <= 8) return launch_reg(tri_eig_kernel<2, 512, 8>);
      return launch_reg(tri_eig_kernel<2, 512, 16>);
    } else {
      return launch_reg(tri_eig_kernel<4, 512, 25>);       // 129 .. 200 rows
    }
  }
  VIPMI_REQUIRE(gat.G == nullptr, "eigh_topk: the gathered input needs the register-resident solver (n = %d, k = %d)", n, k);
  const size_t lds = ((size_t)(9 + nt / 64) * n + 64 + 8) * sizeof(double);     // + prow[n], pcolw[waves][n]
  const void* kern = nt == 256   ? reinterpret_cast<const void*>(tri_eig_kernel<RPL, 256>)
                     : nt == 512 ? reinterpret_cast<const void*>(tri_eig_kernel<RPL, 512>)
                                 : reinterpret_cast<const void*>(tri_eig_kernel<RPL, 1024>);
  VIPMI_CHECK_HIP(set_dyn_lds(kern, (int)lds));
  if (nt == 256)
    hipLaunchKernelGGL((tri_eig_kernel<RPL, 256>), dim3((unsigned)batch), dim3(256), lds, ctx->stream, A, n, k, nact,
                       evals, evecs, scratch, kp, all_evals, k < n ? k : n);
  else if (nt == 512)
    hipLaunchKernelGGL((tri_eig_kernel<RPL, 512>), dim3((unsigned)batch), dim3(512), lds, ctx->stream, A, n, k, nact,
                       evals, evecs, scratch, kp, all_evals, k < n ? k : n);
  else
    hipLaunchKernelGGL((tri_eig_kernel<RPL, 1024>), dim3((unsigned)batch), dim3(1024), lds, ctx->stream, A, n, k, nact,
                       evals, evecs, scratch, kp, all_evals, k < n ? k : n);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

}  // namespace

bool eigh_topk_supported(int64_t n, int64_t k) { return n >= 1 && n <= 512 && k >= 1 && k <= 64; }

// Leading k eigenpairs of `batch` symmetric n x n matrices (destroyed).  evals[p*n + c], evecs[p*n*n + c*n + i] for
// c < k (other entries are not written).  nact (device, optional): active leading size of every problem.
int eigh_topk_f64(vipmi_ctx* ctx, double* A, int64_t batch, int64_t n, int64_t k, const int32_t* nact, double* evals,
                  double* evecs, bool all_evals) {
  VIPMI_REQUIRE(A && evals && evecs, "eigh_topk: null pointer");
  VIPMI_REQUIRE(batch > 0 && eigh_topk_supported(n, k), "eigh_topk: unsupported sizes n=%ld k=%ld", (long)n, (long)k);
  StageScope sc(ctx, "eigh");
  // a few larger problems: spread each over several CUs (LDS-resident matrix); many problems: one CU each
  // (up to 200 rows the register-resident single-workgroup variant is the faster one even for a lone problem:
  // 0.31 / 0.82 ms against 0.42 / 0.93 ms at n = 100 / 200)
  const bool reg1 = ctx->opt("eigh_reg", 1) != 0 && reg_variant_fits((int)n, (int)k);
  // (a lone synchronous problem of 129 .. 200 rows: the wave-resident path of launch_tri_multi beats the register-resident
  //  single-workgroup kernel as well -- n = 200, k = 10: 0.82 -> see tools/eigh_wave_check.py)
  const bool wave1 = batch == 1 && !nact && tri_wave_fits(ctx, n) && ctx->opt("eigh_wave", 1) != 0 && (ctx->opt("eigh_check", 1) != 0 || ctx->opt("eigh_wave_async", 0) != 0) &&
                     ctx->opt("eigh_one_xcd", -1) != 0 && ctx->num_cu % 8 == 0 && ctx->num_cu >= 64;
  const bool multi = !nact && n >= 96 && batch <= 8 && (!reg1 || wave1) && ctx->opt("eigh_multi", 1) != 0;
  if (multi) {
    if (n <= 128) return launch_tri_multi<2>(ctx, A, batch, (int)n, (int)k, evals, evecs, all_evals);
    if (n <= 256) return launch_tri_multi<4>(ctx, A, batch, (int)n, (int)k, evals, evecs, all_evals);
    return launch_tri_multi<8>(ctx, A, batch, (int)n, (int)k, evals, evecs, all_evals);
  }
  if (n <= 128) return launch_tri<2>(ctx, A, batch, (int)n, (int)k, nact, evals, evecs, all_evals);
  if (n <= 256) return launch_tri<4>(ctx, A, batch, (int)n, (int)k, nact, evals, evecs, all_evals);
  return launch_tri<8>(ctx, A, batch, (int)n, (int)k, nact, evals, evecs, all_evals);
}

// Leading k eigenpairs of nseg * per_seg zero-padded library sub-Gram matrices (m x m) of annular PCA, gathered by the solver
// itself from the segments' Gram matrices G[nseg][ldg][ldg]: problem p uses rows / columns idx[p][0 .. len[p]) of G[p / per_seg].
// work[p][m][m]: workspace for the reflectors.  Returns false when the register-resident solver does not serve (m, k) -- the caller
// then materialises the matrices and calls eigh_leading.
bool eigh_gather_supported(int64_t m, int64_t k) {
  return m >= 1 && m <= 200 && k >= 1 && eigh_topk_supported(m, k) && reg_variant_fits((int)m, (int)k);
}
int eigh_topk_gather_f64(vipmi_ctx* ctx, const double* G, int64_t nseg, int64_t per_seg, int64_t ldg, const int32_t* idx,
                         const int32_t* len, int64_t m, int64_t k, double* work, double* evals, double* evecs) {
  VIPMI_REQUIRE(G && idx && len && work && evals && evecs, "eigh_topk_gather: null pointer");
  VIPMI_REQUIRE(eigh_gather_supported(m, k) && ctx->opt("eigh_reg", 1) != 0, "eigh_topk_gather: unsupported sizes m=%ld k=%ld", (long)m, (long)k);
  StageScope sc(ctx, "eigh");
  TriGather gat;
  gat.G = G;
  gat.idx = idx;
  gat.ldg = (int)ldg;
  gat.stride = (int)m;
  gat.per_seg = (int)per_seg;
  const int64_t batch = nseg * per_seg;
  if (m <= 128) return launch_tri<2>(ctx, work, batch, (int)m, (int)k, len, evals, evecs, false, gat);
  return launch_tri<4>(ctx, work, batch, (int)m, (int)k, len, evals, evecs, false, gat);
}

int eigh_leading(vipmi_ctx* ctx, double* G, int64_t batch, int64_t n, int64_t k, const int32_t* nact, double* evals,
                 double* evecs, bool all_evals) {
  // One larger problem, leading pairs only, synchronous mode (the fast path reads Ritz values back every round): the
  // verified Chebyshev-filtered subspace iteration first (eigh_chfsi.hip); it leaves G untouched, so when the spectrum does
  // not allow it within its budget the exact path below runs as if nothing had happened.
  if (batch == 1 && !nact && !all_evals && ctx->opt("eigh_fast", 1) != 0 && ctx->opt("eigh_check", 1) != 0 &&
      ctx->opt("eigh_method", 0) != 1 &&
      n >= (ctx->opt("eigh_fast_min", 0) > 0 ? ctx->opt("eigh_fast_min", 0) : eigh_chfsi_pays_from(k)) && eigh_chfsi_supported(n, k)) {
    int conv = 0, info[4];
    {
      StageScope sc(ctx, "eigh");
      VIPMI_TRY(eigh_chfsi_f64(ctx, G, n, k, evals, evecs, &conv, info));
    }
    ctx->options["eigh_fast_last_products"] = info[0];
    ctx->options["eigh_fast_last_rounds"] = info[1];
    ctx->options["eigh_fast_last_locked"] = info[2];
    ctx->options["eigh_fast_last_reason"] = info[3];
    if (conv) return VIPMI_OK;
  }
  // MANY vectors of one matrix (pca(ncomp = 200), a float ncomp whose CEVR asks for most of the spectrum): the tridiagonal solvers
  // take their vectors 64 at a time, one back-transformation of n reflectors per vector -- 16.6 ms for 200 of 400, 53 ms for all 400,
  // 357 ms for 800 of 800 --, while the one-sided Jacobi kernel delivers every eigenpair in ~10 ms at n = 400, 22-50 ms at n = 800
  // (tools/time_manyvec.py; residuals 5e-14 of the largest eigenvalue against 3e-16).  From k > 0.4 n on (synchronous mode: the
  // convergence flag is read back) Jacobi goes first, on a copy: graded spectra beyond ~600 rows may not converge within its sweep
  // limit, and the tridiagonal path then runs on the untouched matrix.
  if (batch == 1 && !nact && n <= 2048 && k > 64 && 5 * k > 2 * n && ctx->opt("eigh_check", 1) != 0 && ctx->opt("eigh_method", 0) == 0 &&
      ctx->opt("eigh_many_jacobi", 1) != 0) {
    double* Gc = nullptr;
    VIPMI_TRY(ws(ctx, "eigh_jacobi_copy", (size_t)n * n, &Gc));
    VIPMI_CHECK_HIP(hipMemcpyAsync(Gc, G, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, ctx->stream));
    const int st = eigh_f64(ctx, Gc, 1, n, evals, evecs);
    if (st == VIPMI_OK) return VIPMI_OK;
    if (st != VIPMI_ERR_NOCONV) return st;
    set_error("");                                  // (not converged: the exact path below)
    // the kernel also latched the failure for vipmi_check_deferred (sticky word 0): taken back, the caller gets its eigenpairs
    {
      int* fail = nullptr;
      VIPMI_TRY(deferred_fail_words(ctx, &fail, false));
      int v = 0;
      VIPMI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      VIPMI_CHECK_HIP(hipMemcpy(&v, fail, sizeof(int), hipMemcpyDeviceToHost));
      if (v > 0) {
        v -= 1;
        VIPMI_CHECK_HIP(hipMemcpy(fail, &v, sizeof(int), hipMemcpyHostToDevice));
      }
    }
  }
  if (ctx->opt("eigh_method", 0) != 1 && eigh_topk_supported(n, k))
    return eigh_topk_f64(ctx, G, batch, n, k, nact, evals, evecs, all_evals);
  // 513 .. 640 rows, a few problems: still LDS-resident on 32 workgroups (20 rows of 640 doubles + ten vectors = 154 KB each) --
  // 3.1 ms at n = 640 (one XCD; 3.7 ms spread) against 5.7 ms for the matrix-in-L2 kernel below
  if (ctx->opt("eigh_method", 0) != 1 && !nact && batch <= 8 && n > 512 && n <= 640 && k >= 1 && k <= 64 &&
      ctx->opt("eigh_multi", 1) != 0 && ctx->num_cu >= 32) {
    StageScope sc(ctx, "eigh");
    return launch_tri_multi<10>(ctx, G, batch, (int)n, (int)k, evals, evecs, all_evals);
  }
  if (ctx->opt("eigh_method", 0) != 1 && !nact && batch <= 4 && eigh_large_supported(n, k))
    return eigh_large_f64(ctx, G, batch, n, k, evals, evecs, all_evals);
  // Bigger batches that the matrix-in-L2 solver could serve one problem at a time (more than 64 vectors per matrix: a 4-D cube with
  // ncomp = 100) take the one-sided Jacobi kernel -- every problem at once -- and, in synchronous mode, fall back on that solver for
  // the problems Jacobi gives up on (a spectrum graded over twelve decades: round 6) instead of failing the call.
  if (ctx->opt("eigh_method", 0) == 0 && !nact && batch > 4 && eigh_large_supported(n, k) && ctx->opt("eigh_check", 1) != 0) {
    double* Gc = nullptr;
    if (ws(ctx, "eigh_batch_copy", (size_t)batch * n * n, &Gc) != VIPMI_OK) {      // (no room for the copy: the plain call, as before)
      set_error("");
      (void)hipGetLastError();                          // (the failed hipMalloc is HIP's "last error" otherwise)
      return eigh_f64(ctx, G, batch, n, evals, evecs);
    }
    VIPMI_CHECK_HIP(hipMemcpyAsync(Gc, G, sizeof(double) * (size_t)batch * n * n, hipMemcpyDeviceToDevice, ctx->stream));
    const int st = eigh_f64(ctx, G, batch, n, evals, evecs);
    if (st != VIPMI_ERR_NOCONV) return st;
    int* info = nullptr;
    VIPMI_TRY(ws(ctx, "eigh_info", (size_t)batch, &info));             // (sweeps per problem of the launch above; < 0: gave up)
    std::vector<int> h(batch);
    VIPMI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    VIPMI_CHECK_HIP(hipMemcpy(h.data(), info, sizeof(int) * batch, hipMemcpyDeviceToHost));
    int nf = 0;
    for (int64_t p = 0; p < batch; ++p) nf += h[p] < 0 ? 1 : 0;
    set_error("");
    {                                                                   // the latched failures are taken back: the caller gets its eigenpairs
      int* fail = nullptr;
      VIPMI_TRY(deferred_fail_words(ctx, &fail, false));
      int v = 0;
      VIPMI_CHECK_HIP(hipMemcpy(&v, fail, sizeof(int), hipMemcpyDeviceToHost));
      v = v > nf ? v - nf : 0;
      VIPMI_CHECK_HIP(hipMemcpy(fail, &v, sizeof(int), hipMemcpyHostToDevice));
    }
    for (int64_t p = 0; p < batch; ++p)
      if (h[p] < 0)
        VIPMI_TRY(eigh_large_f64(ctx, Gc + (size_t)p * n * n, 1, n, k, evals + (size_t)p * n, evecs + (size_t)p * n * n, all_evals));
    ctx->options["eigh_batch_fallback"] = nf;
    return VIPMI_OK;
  }
  return eigh_f64(ctx, G, batch, n, evals, evecs);
}

}  // namespace vipmi
