// eigh_chfsi.hip -- fast path for the leading k eigenpairs of ONE symmetric positive semi-definite float64 matrix (the Gram
// matrix of psfsub/svd.py:447-491): Chebyshev-filtered block subspace iteration with locking, VERIFIED -- every returned pair
// satisfies ||G q - theta q|| <= tol * theta_1 -- and abandoned (the caller then runs the exact tridiagonal path on the
// untouched matrix) when the spectrum does not allow it within a budget of block products.
//
// Why: the exact path is a chain of n dependent Householder steps (2.1 ms at n = 400, 31 ms at n = 2000: two fabric round
// trips per step, eigh_tri.hip); everything here is a block product Y = G X on the float64 matrix cores
// (v_mfma_f64_16x16x4_f64) plus O(n b^2) dense work on a block of b = k + 12 .. 64 vectors.
//
// Algorithm (host-driven; the host reads b Ritz values + residual norms per outer iteration and takes the scalar decisions):
//   X <- G * (fixed pseudo-random block);  Rayleigh-Ritz
//   repeat
//     lock the leading Ritz pairs whose residual is below tol * theta_1 (at most k), deflate them out of a copy of G
//     unwanted interval [0, bb], bb = smallest Ritz value of the block (G is positive semi-definite: 0 is a safe lower end)
//     degree m limited by the dynamic range the filter may create inside the block: T_m(x_top) <= 2e7 (= against T_m(1))
//     Y <- T_m((Gd - c) / e) Q   by the scaled three-term recurrence (one launch per degree)
//     Y <- Y - L (L^T Y);  Cholesky-QR twice;  H = Q^T G Q;  Jacobi;  Q <- Q W;  residual norms
//   until k pairs are locked, or the forecast of the remaining products exceeds the budget (-> not converged)
// The BASELINE generator puts the k = 20 (C2) / 50 (C5) boundary inside the noise bulk (lambda_33 / lambda_20 = 0.992 at
// C2): plain subspace iteration would need ~3000 products there, the Chebyshev filter ~190 (C2) / ~390 (C5).
#include "common.h"
#include <algorithm>
#include <cmath>
#include <random>

namespace vipmi {

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int BMAX = 64;            // widest block (columns), also the leading dimension of every b x b matrix

// ---- Y = alpha (G X - cshift X) + beta Zp;  G symmetric n x n;  X, Zp, Y: [n][LDB] ------------------------------------------
// One workgroup per 16-row tile and group of NT column tiles (blockIdx.y), NW waves split the contraction, partial tiles
// meet in LDS (fixed order: deterministic).  Operands of KU contraction steps are requested before their MFMAs are issued:
// the loop is a chain of L2 round trips otherwise (18 us per product at n = 400 instead of 3).
template <int NT, int NW, int LDB>
__global__ __launch_bounds__(64 * NW) void cheb_step_kernel(const double* __restrict__ G, int n,
                                                            const double* __restrict__ X,
                                                            const double* __restrict__ Zp, double* __restrict__ Y,
                                                            double alpha, double cshift, double beta) {
  extern __shared__ double lds[];                 // [NW][NT][256]
  constexpr int KU = 8;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r0 = blockIdx.x * 16, c0 = blockIdx.y * 16 * NT;
  const int r = lane & 15, kq = lane >> 4;
  const int ksteps = (n + 3) >> 2;
  const int per = (ksteps + NW - 1) / NW;
  const int ks0 = wave * per, ks1 = min(ksteps, ks0 + per);
  d4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
  const bool rok = r0 + r < n;
  const double* gcol = G + r0 + r;                // A[row r][k] = G[r0 + r][k] = G[k][r0 + r]: 128 contiguous bytes per k
  const double* xcol = X + c0 + r;
  // two register sets of KU contraction steps each: the loads of one set are in flight while the MFMAs of the other issue
  double a0[KU], b0[KU][NT], a1[KU], b1[KU][NT];
  auto load_set = [&](int ks, double (&a)[KU], double (&bv)[KU][NT]) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int kk = (ks + u) * 4 + kq;
      const bool kok = (ks + u < ks1) && kk < n;
      a[u] = (kok && rok) ? gcol[(size_t)kk * n] : 0.0;
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[u][t] = kok ? xcol[(size_t)kk * LDB + 16 * t] : 0.0;
    }
  };
  auto mma_set = [&](const double (&a)[KU], const double (&bv)[KU][NT]) {
#pragma unroll
    for (int u = 0; u < KU; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], bv[u][t], acc[t], 0, 0, 0);
  };
  load_set(ks0, a0, b0);
  for (int ks = ks0; ks < ks1; ks += 2 * KU) {
    load_set(ks + KU, a1, b1);          // (past the end: zeros, the MFMAs below then add nothing)
    mma_set(a0, b0);
    load_set(ks + 2 * KU, a0, b0);
    mma_set(a1, b1);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) lds[(wave * NT + t) * 256 + (kq + 4 * g) * 16 + r] = acc[t][g];
  __syncthreads();
  for (int idx = threadIdx.x; idx < 256 * NT; idx += 64 * NW) {
    const int row = idx / (16 * NT), cl = idx % (16 * NT);
    const int grow = r0 + row;
    if (grow >= n) continue;
    const int t = cl >> 4, c = cl & 15;
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += lds[(w * NT + t) * 256 + row * 16 + c];
    const size_t o = (size_t)grow * LDB + c0 + cl;
    double v = alpha * (s - cshift * X[o]);
    if (Zp) v += beta * Zp[o];
    Y[o] = v;
  }
}

// ---- part[s][i][j] = sum over the rows of chunk s of A[r][i] B[r][j]  (i < p, j < q <= 64) ------------------------------------
__global__ __launch_bounds__(256) void tn_partial_kernel(const double* __restrict__ A, int lda, int p,
                                                         const double* __restrict__ B, int ldb, int q, int n,
                                                         int rows_per, double* __restrict__ part) {
  constexpr int KC = 8;
  __shared__ double As[KC][BMAX], Bs[KC][BMAX];
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  const int ra = blockIdx.x * rows_per, rb = min(n, ra + rows_per);
  double acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
  for (int k0 = ra; k0 < rb; k0 += KC) {
    for (int e = tid; e < KC * BMAX; e += 256) {
      const int kk = e / BMAX, c = e % BMAX;
      const int row = k0 + kk;
      As[kk][c] = (row < rb && c < p) ? A[(size_t)row * lda + c] : 0.0;
      Bs[kk][c] = (row < rb && c < q) ? B[(size_t)row * ldb + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      double a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = As[kk][ti + 16 * u];
#pragma unroll
      for (int v = 0; v < 4; ++v) b[v] = Bs[kk][tj + 16 * v];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = fma(a[u], b[v], acc[u][v]);
    }
    __syncthreads();
  }
  double* out = part + (size_t)blockIdx.x * BMAX * BMAX;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) out[(ti + 16 * u) * BMAX + tj + 16 * v] = acc[u][v];
}

// sum of the partial products of tn_partial_kernel into a [64][LD] LDS matrix (fixed order)
template <int LD>
__device__ __forceinline__ void load_partials(double (*M)[LD], const double* __restrict__ part, int nsplit, int p, int q) {
  for (int e = threadIdx.x; e < BMAX * BMAX; e += blockDim.x) {
    const int i = e / BMAX, j = e % BMAX;
    double s = 0.0;
    if (i < p && j < q)
      for (int sp = 0; sp < nsplit; ++sp) s += part[(size_t)sp * BMAX * BMAX + e];
    M[i][j] = s;
  }
}

// ---- S = sum of partials (p x p);  S = L L^T;  Rinv = L^-T (upper)  ->  Y Rinv has orthonormal columns ---------------------------
// status[0] |= 1 when a pivot is not positive (the block has lost rank: the caller gives up).  1024 threads: the trailing
// update of a column step is one element per thread, and the inverse is built a column per wave (the forward substitution of
// a column is sequential in the row, its inner product runs over the lanes).
__global__ __launch_bounds__(1024) void chol_inv_kernel(const double* __restrict__ part, int nsplit, int p,
                                                        double* __restrict__ Rinv, int* __restrict__ status) {
  __shared__ double S[BMAX][BMAX + 1], Li[BMAX][BMAX + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int far_;
  __shared__ double rdiag[BMAX];
  load_partials(S, part, nsplit, p, p);
  for (int e = tid; e < BMAX * (BMAX + 1); e += 1024) (&Li[0][0])[e] = 0.0;
  if (tid == 0) far_ = 0;
  __syncthreads();
  // second pass of a well conditioned block: Y^T Y is the identity to round-off already -> nothing to factor
  for (int e = tid; e < p * p; e += 1024) {
    const int i = e / p, j = e % p;
    const double dlt = fabs(S[i][j] - (i == j ? 1.0 : 0.0));
    if (!(dlt < 2e-14)) far_ = 1;
  }
  __syncthreads();
  if (!far_) {
    for (int e = tid; e < BMAX * BMAX; e += 1024) Rinv[e] = (e / BMAX == e % BMAX && e / BMAX < p) ? 1.0 : 0.0;
    return;
  }
  // right-looking, ONE barrier per column: the trailing update divides by the pivot itself, column j keeps its unscaled
  // entries (nothing touches them after step j) and is scaled by 1 / sqrt(pivot) once at the end
  bool bad = false;
  for (int j = 0; j < p; ++j) {
    const double d = S[j][j];
    if (!(d > 0.0)) {                       // (also catches NaN; uniform: every thread reads the same value)
      bad = true;
      break;
    }
    const double rd = 1.0 / d;
    const int m = p - j - 1;                // rows i > j, columns j < c <= i
    for (int e = tid; e < m * m; e += 1024) {
      const int i = j + 1 + e / m, c = j + 1 + e % m;
      if (c <= i) S[i][c] -= S[i][j] * S[c][j] * rd;
    }
    __syncthreads();
  }
  if (!bad) {
    for (int e = tid; e < p * p; e += 1024) {
      const int i = e / p, j = e % p;
      if (i > j) S[i][j] = S[i][j] / sqrt(S[j][j]);
    }
    __syncthreads();
    if (tid < p) {
      const double l = sqrt(S[tid][tid]);
      S[tid][tid] = l;
      rdiag[tid] = 1.0 / l;
    }
    __syncthreads();
  }
  if (bad) {
    if (tid == 0) atomicOr(status, 1);
    for (int e = tid; e < BMAX * BMAX; e += 1024) Rinv[e] = 0.0;
    return;
  }
  // Li = L^-1 (lower): wave w builds columns w, w + 16, ...; x_i = -(sum_{c <= k < i} L[i][k] x_k) / L[i][i]
  for (int c = wave; c < p; c += 16) {
    double xk = (lane == c) ? rdiag[c] : 0.0;              // lane k holds x_k
    for (int i = c + 1; i < p; ++i) {
      double t = (lane >= c && lane < i) ? S[i][lane] * xk : 0.0;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
      if (lane == i) xk = -t * rdiag[i];
    }
    if (lane < p) Li[lane][c] = xk;
  }
  __syncthreads();
  for (int e = tid; e < BMAX * BMAX; e += 1024) {
    const int i = e / BMAX, j = e % BMAX;
    Rinv[e] = (i < p && j < p && i <= j) ? Li[j][i] : 0.0;      // Rinv = Li^T
  }
}

// ---- H = sym(sum of partials) (p x p);  cyclic two-sided Jacobi;  W columns = eigenvectors, theta descending -----------------------
__global__ __launch_bounds__(1024) void jacobi_small_kernel(const double* __restrict__ part, int nsplit, int p,
                                                           double* __restrict__ W, double* __restrict__ theta) {
  __shared__ double H[BMAX][BMAX + 1], V[BMAX][BMAX + 1];
  __shared__ double cs[BMAX / 2], sn[BMAX / 2];
  __shared__ int pa[BMAX / 2], pb[BMAX / 2];
  __shared__ int rank[BMAX];
  const int tid = threadIdx.x, nthr = blockDim.x;        // 256 threads up to 32 columns (cheaper barriers), 1024 above
  load_partials(H, part, nsplit, p, p);
  __syncthreads();
  for (int e = tid; e < BMAX * BMAX; e += nthr) {
    const int i = e / BMAX, j = e % BMAX;
    V[i][j] = (i == j) ? 1.0 : 0.0;
    if (i < j && j < p) {                          // symmetrise (the two halves differ in round-off)
      const double s = 0.5 * (H[i][j] + H[j][i]);
      H[i][j] = s;
      H[j][i] = s;
    }
  }
  __syncthreads();
  const int m = p + (p & 1);                       // even number of players; index p (if any) is a bye
  const int half = m >> 1;
  __shared__ int rotated, finite, round_rot[2];
  if (tid == 0) finite = 1;
  __syncthreads();
  for (int e = tid; e < p * p; e += nthr) {
    const double h = H[e / p][e % p];
    if (!(h == h) || fabs(h) > 1e300) finite = 0;
  }
  __syncthreads();
  // threshold Jacobi: a pair is rotated only while |h_ab| > 1e-15 sqrt(|h_aa h_bb|) (+ an absolute floor against the largest
  // diagonal entry); the sweeps end with the first one that rotates nothing.  The block is a filtered set of the previous Ritz
  // vectors, so H is close to diagonal after the first rounds and two or three sweeps do.
  double dmax = 0.0;
  for (int i = 0; i < p; ++i) dmax = fmax(dmax, fabs(H[i][i]));
  const double idm = dmax > 0.0 ? 1.0 / dmax : 0.0;
  const double floor2 = (2e-15 * dmax) * (2e-15 * dmax);          // (round-off of H itself is a few 1e-16 dmax: rotating it away never ends)
  for (int sweep = 0; sweep < 14 && finite; ++sweep) {
    if (tid == 0) { rotated = 0; round_rot[0] = 0; round_rot[1] = 0; }
    __syncthreads();
    for (int rd = 0; rd < m - 1; ++rd) {
      int* rr_flag = &round_rot[rd & 1];
      // the slot of the NEXT round is cleared here: its last readers are behind a barrier (or read the 0 it already held),
      // its next writers come after this round's first barrier
      if (tid == 0) round_rot[(rd + 1) & 1] = 0;
      if (tid < half) {
        int a, b;
        if (tid == 0) { a = m - 1; b = rd; }
        else { a = (rd + tid) % (m - 1); b = (rd - tid + (m - 1)) % (m - 1); }
        if (a > b) { const int t_ = a; a = b; b = t_; }
        double c = 1.0, s = 0.0;
        if (b < p) {
          const double apq = H[a][b], app = H[a][a], aqq = H[b][b];
          if (apq * apq > 1e-30 * fabs(app * aqq) + floor2) {
            // the tangent only steers the convergence: float32 arithmetic (a float64 division and two square roots are
            // ~1 us of dependent instructions per round); c = (1 + t^2)^-1/2 is refined to float64 so that the rotation is
            // orthogonal to round-off whatever t is
            const float tau = (float)((aqq - app) * idm) / (2.f * (float)(apq * idm));     // (scaled: no float under/overflow)
            const float tf = (tau >= 0.f ? 1.f : -1.f) / (fabsf(tau) + sqrtf(1.f + tau * tau));
            const double t = (double)tf, x = 1.0 + t * t;
            double y = (double)rsqrtf((float)x);
            y = y * (1.5 - 0.5 * x * y * y);
            y = y * (1.5 - 0.5 * x * y * y);
            c = y;
            s = t * c;
            rotated = 1;
            *rr_flag = 1;
          }
        }
        pa[tid] = a; pb[tid] = b; cs[tid] = c; sn[tid] = s;
      }
      __syncthreads();
      const bool any = *rr_flag != 0;                      // (uniform) a round without rotations costs one barrier, not three
      if (!any) continue;
      // H <- J^T H J by 2 x 2 blocks (rows of pair P1, columns of pair P2): every block is independent, one per thread
      if (tid < half * half) {
        const int P1 = tid / half, P2 = tid % half;
        const double s1 = sn[P1], s2 = sn[P2];
        if (s1 != 0.0 || s2 != 0.0) {
          const double c1 = cs[P1], c2 = cs[P2];
          const int a1 = pa[P1], b1 = pb[P1], a2 = pa[P2], b2 = pb[P2];
          const double haa = H[a1][a2], hab = H[a1][b2], hba = H[b1][a2], hbb = H[b1][b2];
          const double ta = c2 * haa - s2 * hab, tb = s2 * haa + c2 * hab;      // row a1, columns (a2, b2)
          const double ua = c2 * hba - s2 * hbb, ub = s2 * hba + c2 * hbb;      // row b1
          H[a1][a2] = c1 * ta - s1 * ua;
          H[a1][b2] = c1 * tb - s1 * ub;
          H[b1][a2] = s1 * ta + c1 * ua;
          H[b1][b2] = s1 * tb + c1 * ub;
        }
      }
      for (int e = tid; e < half * BMAX; e += nthr) {       // columns a, b of V
        const int q = e / BMAX, i = e % BMAX;
        const double s_ = sn[q];
        if (s_ == 0.0 || i >= p) continue;
        const double c_ = cs[q];
        const int a = pa[q], b = pb[q];
        const double va = V[i][a], vb = V[i][b];
        V[i][a] = c_ * va - s_ * vb;
        V[i][b] = s_ * va + c_ * vb;
      }
      __syncthreads();
    }
    __syncthreads();                               // every thread is past the last round before the flag is read ...
    const int again = rotated;
    __syncthreads();                               // ... and has read it before the next sweep clears it
    if (!again) break;
  }
  if (tid < p) {
    const double d = H[tid][tid];
    int rk = 0;
    for (int i = 0; i < p; ++i) {
      const double di = H[i][i];
      rk += (di > d || (di == d && i < tid)) ? 1 : 0;
    }
    rank[tid] = rk;
    theta[rk] = finite ? d : __longlong_as_double(0x7ff8000000000000ll);
  }
  for (int e = tid; e < BMAX; e += nthr) if (e >= p) theta[e] = 0.0;
  __syncthreads();
  for (int e = tid; e < BMAX * BMAX; e += nthr) W[e] = 0.0;
  __syncthreads();
  for (int e = tid; e < p * p; e += nthr) {
    const int i = e / p, j = e % p;
    W[i * BMAX + rank[j]] = V[i][j];
  }
}

// ---- Out[r][j] = (Base ? Base[r][j] - : ) sum_i In[r][i] M[i][j],  j < q (columns q .. ldo-1 are zeroed) ------------------------
// M = sum of `nsplit` [64][64] blocks (nsplit = 1: a plain matrix).  With In2 / Out2 the same product is formed for a second
// block and rpart[block][j] = sum over the block's rows of (Out2[r][j] - theta[j] Out[r][j])^2 (residual norms, RR step).
__global__ __launch_bounds__(256) void nn_small_kernel(const double* __restrict__ In, int ldi, int p,
                                                       const double* __restrict__ Mg, int nsplit, int q,
                                                       const double* __restrict__ Base, double* __restrict__ Out, int ldo,
                                                       int n, const double* __restrict__ In2, double* __restrict__ Out2,
                                                       const double* __restrict__ theta, double* __restrict__ rpart) {
  __shared__ double M[BMAX][BMAX + 1];
  __shared__ double rs[16][BMAX];
  load_partials(M, Mg, nsplit, p, q);
  __syncthreads();
  const int tid = threadIdx.x, rr = tid >> 4, jc = tid & 15;
  const int r = blockIdx.x * 16 + rr;
  double o[4] = {0.0, 0.0, 0.0, 0.0}, o2[4] = {0.0, 0.0, 0.0, 0.0};
  if (r < n) {
    const double* x = In + (size_t)r * ldi;
    const double* x2 = In2 ? In2 + (size_t)r * ldi : nullptr;
    for (int i = 0; i < p; ++i) {
      const double xi = x[i];
#pragma unroll
      for (int v = 0; v < 4; ++v) o[v] = fma(xi, M[i][jc + 16 * v], o[v]);
      if (x2) {
        const double yi = x2[i];
#pragma unroll
        for (int v = 0; v < 4; ++v) o2[v] = fma(yi, M[i][jc + 16 * v], o2[v]);
      }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int j = jc + 16 * v;
      if (j >= ldo) continue;
      double val = (j < q) ? o[v] : 0.0;
      if (Base && j < q) val = Base[(size_t)r * ldo + j] - val;
      Out[(size_t)r * ldo + j] = val;
      if (Out2) Out2[(size_t)r * ldo + j] = (j < q) ? o2[v] : 0.0;
    }
  }
  if (rpart) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int j = jc + 16 * v;
      const double d = (r < n && j < q) ? (o2[v] - theta[j] * o[v]) : 0.0;
      rs[rr][j] = d * d;
    }
    __syncthreads();
    if (tid < BMAX) {
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < 16; ++a) s += rs[a][tid];
      rpart[(size_t)blockIdx.x * BMAX + tid] = s;
    }
  }
}

// ---- lock the nl leading columns of Q (append to L, deflate out of Gd) and close the gap in Q -----------------------------------
__global__ void lock_copy_kernel(const double* __restrict__ Q, int ldb, int n, int nl, int bact, double* __restrict__ L,
                                 int ldl, int nlock, double* __restrict__ Qn) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * ldb) return;
  const int r = idx / ldb, j = idx % ldb;
  if (j < nl) L[(size_t)r * ldl + nlock + j] = Q[idx];
  Qn[idx] = (j + nl < bact) ? Q[(size_t)r * ldb + j + nl] : 0.0;
}

__global__ void deflate_kernel(double* __restrict__ Gd, int n, const double* __restrict__ L, int ldl, int c0, int nl,
                               const double* __restrict__ lth) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= n) return;
  double s = 0.0;
  for (int l = 0; l < nl; ++l) s = fma(lth[c0 + l] * L[(size_t)i * ldl + c0 + l], L[(size_t)j * ldl + c0 + l], s);
  Gd[(size_t)i * n + j] -= s;
}

// ---- output: row c of evecs = locked vector perm[c], sign convention of eigh.hip (largest |component| positive) ----------------
__global__ __launch_bounds__(256) void finalize_kernel(const double* __restrict__ L, int ldl, int n, const int* __restrict__ perm,
                                                       const double* __restrict__ lth, double* __restrict__ evals,
                                                       double* __restrict__ evecs) {
  __shared__ double bv[256], bs[256];
  __shared__ int bi[256];
  const int c = blockIdx.x, src = perm[c], tid = threadIdx.x;
  double best = -1.0, bval = 0.0;
  int bidx = 0x7fffffff;
  for (int i = tid; i < n; i += 256) {
    const double v = L[(size_t)i * ldl + src], a = fabs(v);
    if (a > best) { best = a; bval = v; bidx = i; }
  }
  bv[tid] = best; bs[tid] = bval; bi[tid] = bidx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      if (bv[tid + s] > bv[tid] || (bv[tid + s] == bv[tid] && bi[tid + s] < bi[tid])) {
        bv[tid] = bv[tid + s]; bs[tid] = bs[tid + s]; bi[tid] = bi[tid + s];
      }
    }
    __syncthreads();
  }
  const double sg = bs[0] < 0.0 ? -1.0 : 1.0;
  for (int i = tid; i < n; i += 256) evecs[(size_t)c * n + i] = sg * L[(size_t)i * ldl + src];
  if (tid == 0) evals[c] = lth[src];
}

struct Chfsi {
  vipmi_ctx* ctx;
  const double* G;
  int n, k, b, NT, nsplit, rows_per, nrb;
  double *Gd, *X[3], *Z, *T1, *L, *part, *Rinv, *W, *theta_d, *rpart, *lth_d;
  int* status_d;
  double* host;              // pinned: theta[64], rpart[nrb][64], status
  int matvecs = 0, rounds = 0;

  int step(const double* Gm, const double* Xin, const double* Zp, double* Yout, double alpha, double cshift, double beta) {
    // column tiles per workgroup: all of them for small matrices (one pass over G per product), two for large ones
    // (twice the workgroups: 125 row tiles alone leave half of the 256 CUs idle at n = 2000)
    const int ntg = (n > 1024 && NT % 2 == 0) ? 2 : NT;
    const dim3 grid((unsigned)cdiv(n, 16), (unsigned)(NT / ntg));
#define VIPMI_CHF_LAUNCH(NT_, NW_, LDB_)                                                                                \
  hipLaunchKernelGGL((cheb_step_kernel<NT_, NW_, LDB_>), grid, dim3(64 * NW_), (size_t)NW_ * NT_ * 256 * sizeof(double), \
                     ctx->stream, Gm, n, Xin, Zp, Yout, alpha, cshift, beta)
    if (ntg == NT) {
      // (four column tiles: four waves, 32 KB of LDS for the partial tiles -- eight would need all 64 KB of the default limit)
      if (NT == 2) VIPMI_CHF_LAUNCH(2, 8, 32); else if (NT == 3) VIPMI_CHF_LAUNCH(3, 8, 48); else VIPMI_CHF_LAUNCH(4, 4, 64);
    } else {
      if (NT == 2) VIPMI_CHF_LAUNCH(2, 8, 32); else VIPMI_CHF_LAUNCH(2, 8, 64);
    }
#undef VIPMI_CHF_LAUNCH
    VIPMI_CHECK_HIP(hipGetLastError());
    ++matvecs;
    return VIPMI_OK;
  }
  int tn(const double* A, int lda, int p, const double* B, int ldb, int q) {
    hipLaunchKernelGGL(tn_partial_kernel, dim3(nsplit), dim3(256), 0, ctx->stream, A, lda, p, B, ldb, q, n, rows_per, part);
    VIPMI_CHECK_HIP(hipGetLastError());
    return VIPMI_OK;
  }
  int nn(const double* In, int ldi, int p, const double* M, int msplit, int q, const double* Base, double* Out,
         const double* In2 = nullptr, double* Out2 = nullptr, bool resid = false) {
    hipLaunchKernelGGL(nn_small_kernel, dim3(nrb), dim3(256), 0, ctx->stream, In, ldi, p, M, msplit, q, Base, Out, b, n, In2,
                       Out2, resid ? theta_d : nullptr, resid ? rpart : nullptr);
    VIPMI_CHECK_HIP(hipGetLastError());
    return VIPMI_OK;
  }

  // Rayleigh-Ritz on span(Y[:, :bact]) (orthogonalised against the nlock locked vectors): Q (into Qout), theta, residuals.
  // Y is overwritten.  th / res receive the bact Ritz values (descending) and residual norms; *ok = 0 when the block lost rank.
  int rayleigh_ritz(double* Y, double* Qout, int bact, int nlock, double* th, double* res, int* ok) {
    ++rounds;
    const int ldl = BMAX;
    double* cur = Y;
    double* oth = T1;
    if (nlock > 0) {                                 // Y <- Y - L (L^T Y)
      VIPMI_TRY(tn(L, ldl, nlock, cur, b, bact));
      VIPMI_TRY(nn(L, ldl, nlock, part, nsplit, bact, cur, oth));
      std::swap(cur, oth);
    }
    for (int pass = 0; pass < 2; ++pass) {           // Cholesky-QR, twice
      VIPMI_TRY(tn(cur, b, bact, cur, b, bact));
      hipLaunchKernelGGL(chol_inv_kernel, dim3(1), dim3(1024), 0, ctx->stream, part, nsplit, bact, Rinv, status_d);
      VIPMI_CHECK_HIP(hipGetLastError());
      VIPMI_TRY(nn(cur, b, bact, Rinv, 1, bact, nullptr, oth));
      std::swap(cur, oth);
    }
    VIPMI_TRY(step(G, cur, nullptr, Z, 1.0, 0.0, 0.0));              // Z = G Q~   (the ORIGINAL matrix)
    VIPMI_TRY(tn(cur, b, bact, Z, b, bact));                          // H = Q~^T Z
    hipLaunchKernelGGL(jacobi_small_kernel, dim3(1), dim3(bact <= 32 ? 256 : 1024), 0, ctx->stream, part, nsplit, bact, W, theta_d);
    VIPMI_CHECK_HIP(hipGetLastError());
    // Q = Q~ W, ZW = Z W (into `oth`), residual partial sums
    VIPMI_TRY(nn(cur, b, bact, W, 1, bact, nullptr, Qout, Z, oth, true));
    // read back: theta, residual partials, status
    VIPMI_CHECK_HIP(hipMemcpyAsync(host, theta_d, sizeof(double) * BMAX, hipMemcpyDeviceToHost, ctx->stream));
    VIPMI_CHECK_HIP(hipMemcpyAsync(host + BMAX, rpart, sizeof(double) * (size_t)nrb * BMAX, hipMemcpyDeviceToHost, ctx->stream));
    VIPMI_CHECK_HIP(hipMemcpyAsync(host + BMAX + (size_t)nrb * BMAX, status_d, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    VIPMI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    int st = 0;
    memcpy(&st, host + BMAX + (size_t)nrb * BMAX, sizeof(int));
    *ok = (st == 0);
    for (int j = 0; j < bact; ++j) {
      th[j] = host[j];
      double s = 0.0;
      for (int rb = 0; rb < nrb; ++rb) s += host[BMAX + (size_t)rb * BMAX + j];
      res[j] = std::sqrt(s);
      if (!(th[j] == th[j]) || !(res[j] == res[j])) *ok = 0;
    }
    return VIPMI_OK;
  }
};

}  // namespace

bool eigh_chfsi_supported(int64_t n, int64_t k) {
  if (n < 256 || n > 16384 || k < 1) return false;
  const int64_t want = k + std::max<int64_t>(12, k / 4);
  return want <= BMAX && 4 * k <= n;
}

// Smallest matrix on which the fast path is faster than the exact solvers, by block width (tools/eigh_fast_threshold.py, one
// MI355X, wall time incl. the host round trips of the Rayleigh-Ritz rounds): b = 32: 5.5 against 6.2 ms at n = 700 (5.2 / 5.1 at
// 620); b = 48: 7.2 / 7.9 ms at n = 800; b = 64: 10.0 / 6.0 ms at n = 600, 15.7 / 31.9 ms at n = 2000 -- a round costs ~0.3 / 0.45 /
// 0.7 ms at b = 32 / 48 / 64 whatever n is, the exact path grows like n^2 .. n^3.
int64_t eigh_chfsi_pays_from(int64_t k) {
  const int64_t want = k + std::max<int64_t>(12, k / 4);
  return want <= 32 ? 700 : want <= 48 ? 800 : 1000;
}

// Leading k eigenpairs of the symmetric positive semi-definite G (n x n, float64, NOT modified).  *converged = 1: evals[0..k)
// (descending) and evecs (row c = eigenvector c, the sign convention of the exact solvers) are set and every pair satisfies
// ||G q - theta q|| <= tol * theta_1;  *converged = 0: nothing was written, the caller runs the exact path.
// info (optional, 4 ints): block products, Rayleigh-Ritz rounds, locked pairs, reason (0 ok, 1 forecast over budget,
// 2 budget exhausted, 3 rank loss / non-finite values, 4 not positive semi-definite).
int eigh_chfsi_f64(vipmi_ctx* ctx, const double* G, int64_t n64, int64_t k64, double* evals, double* evecs, int* converged,
                   int* info) {
  *converged = 0;
  if (info) info[0] = info[1] = info[2] = info[3] = 0;
  if (!eigh_chfsi_supported(n64, k64)) return VIPMI_OK;
  Chfsi S;
  S.ctx = ctx;
  S.G = G;
  const int n = S.n = (int)n64, k = S.k = (int)k64;
  const int want = k + std::max(12, k / 4);
  // (at least two column tiles: cheb_step_kernel has no one-tile instance -- k <= 4 used to run the 64-column kernel on a
  //  16-column block and fault; the wider block costs nothing at these sizes and converges in fewer rounds)
  const int b = S.b = std::max(32, (int)cdiv(want, 16) * 16);
  S.NT = b / 16;
  S.rows_per = n > 1024 ? 64 : 50;
  S.nsplit = (int)cdiv(n, S.rows_per);
  S.nrb = (int)cdiv(n, 16);
  const double tol = 1e-13 * (double)std::max<int64_t>(1, ctx->opt("eigh_fast_tol", 1));     // residual gate, in units of theta_1
  const double dyn = 2e7;            // largest amplification the filter may create inside the block (T_m(x_top) against T_m(1) = 1)
  const int mmax = 240;
  const int budget = (int)ctx->opt("eigh_fast_budget", n <= 512 ? 260 : 700);
  const size_t nb = (size_t)n * b;
  VIPMI_TRY(ws(ctx, "chf_Gd", (size_t)n * n, &S.Gd));
  for (int i = 0; i < 3; ++i) {
    char nm[16];
    snprintf(nm, sizeof nm, "chf_X%d", i);
    VIPMI_TRY(ws(ctx, nm, nb, &S.X[i]));
  }
  VIPMI_TRY(ws(ctx, "chf_Z", nb, &S.Z));
  VIPMI_TRY(ws(ctx, "chf_T1", nb, &S.T1));
  VIPMI_TRY(ws(ctx, "chf_L", (size_t)n * BMAX, &S.L));
  VIPMI_TRY(ws(ctx, "chf_part", (size_t)S.nsplit * BMAX * BMAX, &S.part));
  VIPMI_TRY(ws(ctx, "chf_Rinv", (size_t)BMAX * BMAX, &S.Rinv));
  VIPMI_TRY(ws(ctx, "chf_W", (size_t)BMAX * BMAX, &S.W));
  VIPMI_TRY(ws(ctx, "chf_theta", (size_t)BMAX, &S.theta_d));
  VIPMI_TRY(ws(ctx, "chf_rpart", (size_t)S.nrb * BMAX, &S.rpart));
  VIPMI_TRY(ws(ctx, "chf_lth", (size_t)BMAX, &S.lth_d));
  VIPMI_TRY(ws(ctx, "chf_status", (size_t)4, &S.status_d));
  int* perm_d = nullptr;
  VIPMI_TRY(ws(ctx, "chf_perm", (size_t)BMAX, &perm_d));
  const size_t host_bytes = sizeof(double) * (BMAX + (size_t)S.nrb * BMAX + 8);
  void* hp = nullptr;
  VIPMI_TRY(ctx->host_scratch(host_bytes, &hp));
  S.host = reinterpret_cast<double*>(hp);
  hipStream_t st = ctx->stream;
  VIPMI_CHECK_HIP(hipMemsetAsync(S.status_d, 0, sizeof(int) * 4, st));

  // fixed pseudo-random start block (uploaded once per (n, b))
  {
    char key[64];
    snprintf(key, sizeof key, "x0:%d:%d", n, b);
    void* x0 = nullptr;
    if (!ctx->cached("chf_rand", key, &x0)) {
      std::vector<double> h(nb);
      std::mt19937_64 rng(0x9e3779b97f4a7c15ull);
      std::normal_distribution<double> nd(0.0, 1.0);
      for (size_t i = 0; i < nb; ++i) h[i] = nd(rng);
      VIPMI_TRY(ctx->upload_cached("chf_rand", key, h.data(), sizeof(double) * nb, &x0));
    }
    VIPMI_TRY(S.step(G, reinterpret_cast<const double*>(x0), nullptr, S.X[1], 1.0, 0.0, 0.0));     // X <- G X0
  }
  double th[BMAX], res[BMAX], lth[BMAX];
  int ok = 1, nlock = 0, bact = b, reason = 0;
  int qi = 0;                                         // X[qi] holds the current Ritz vectors
  VIPMI_TRY(S.rayleigh_ritz(S.X[1], S.X[0], bact, 0, th, res, &ok));
  const double lam1 = th[0];
  if (!ok || !(lam1 > 0.0)) reason = 3;
  for (int it = 0; it < 60 && !reason; ++it) {
    // lock the converged leading pairs
    int nl = 0;
    while (nl < bact && nlock + nl < k && res[nl] <= tol * lam1) ++nl;
    if (nl > 0) {
      for (int j = 0; j < nl; ++j) lth[nlock + j] = th[j];
      VIPMI_CHECK_HIP(hipMemcpyAsync(S.lth_d + nlock, lth + nlock, sizeof(double) * nl, hipMemcpyHostToDevice, st));
      const int other = (qi + 1) % 3;
      hipLaunchKernelGGL(lock_copy_kernel, dim3((unsigned)cdiv((int64_t)n * b, 256)), dim3(256), 0, st, S.X[qi], b, n, nl, bact,
                         S.L, BMAX, nlock, S.X[other]);
      VIPMI_CHECK_HIP(hipGetLastError());
      if (nlock == 0)
        VIPMI_CHECK_HIP(hipMemcpyAsync(S.Gd, G, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(deflate_kernel, dim3((unsigned)cdiv(n, 256), (unsigned)n), dim3(256), 0, st, S.Gd, n, S.L, BMAX, nlock, nl,
                         S.lth_d);
      VIPMI_CHECK_HIP(hipGetLastError());
      // (the pinned lth values must stay put until the copy has run: lth lives on this stack frame for the whole call and is only
      // appended to, and every later read-back synchronises the stream)
      qi = other;
      for (int j = nl; j < bact; ++j) { th[j - nl] = th[j]; res[j - nl] = res[j]; }
      bact -= nl;
      nlock += nl;
    }
    const int kk = k - nlock;
    if (kk <= 0) break;
    if (bact < kk + 2) { reason = 3; break; }
    const double* Gm = nlock ? S.Gd : G;
    // unwanted interval [0, bb]
    const double bb = th[bact - 1];
    if (!(bb > 0.0)) { reason = 4; break; }
    const double c = 0.5 * bb, e = 0.5 * bb;
    const double xt = std::max(1.0, (th[0] - c) / e), xe = std::max(1.0, (th[kk - 1] - c) / e);
    // degree limit: the filter may amplify the top of the block by at most `dyn` over the directions it leaves alone
    // (x = 1: where the padding columns of the block live) -- beyond that the filtered block is rank deficient in float64
    // and the Cholesky-QR loses it (a cluster of k well separated pairs above a gap is exactly that case)
    const double ge = std::acosh(xe), gap = std::acosh(xt);
    int m = (gap > 1e-12) ? (int)std::floor(std::log(dyn) / gap) : mmax;
    m = std::max(1, std::min(mmax, m));
    // degrees still needed for the slowest wanted pair (its error shrinks like 1 / T_m(x_k))
    double rmax = 0.0;
    for (int j = 0; j < kk; ++j) rmax = std::max(rmax, res[j]);
    const double F = std::max(1.0, rmax / (tol * lam1));
    const int need = ge > 1e-9 ? (int)std::ceil((std::log(2.0 * F) + 1.0) / ge) : 1 << 30;
    // forecast (from the second round on, when the Ritz values mean something): give up before the budget is spent on a
    // spectrum without a gap behind the k-th pair; never run a filter that would overshoot the budget
    if (it >= 1 && (double)S.matvecs + (double)need > (double)budget) { reason = 1; break; }
    if (S.matvecs >= budget) { reason = 2; break; }
    m = std::min(m, std::max(2, need));
    if (it < 2) m = std::min(m, 24);
    m = std::min(m, std::max(1, budget - S.matvecs));
    // Y = T_m((Gm - c) / e) Q, scaled three-term recurrence (Zhou & Saad): sigma_1 = e / (theta_top - c)
    double sigma1 = e / (th[0] - c), sigma = sigma1;
    int ip = qi, iy = (qi + 1) % 3, in_ = (qi + 2) % 3;
    VIPMI_TRY(S.step(Gm, S.X[ip], nullptr, S.X[iy], sigma1 / e, c, 0.0));
    for (int j = 2; j <= m; ++j) {
      const double sigma2 = 1.0 / (2.0 / sigma1 - sigma);
      VIPMI_TRY(S.step(Gm, S.X[iy], S.X[ip], S.X[in_], 2.0 * sigma2 / e, c, -sigma * sigma2));
      const int t_ = ip; ip = iy; iy = in_; in_ = t_;
      sigma = sigma2;
    }
    // Rayleigh-Ritz; the Ritz vectors go to a buffer other than the filtered block
    const int iq = (iy + 1) % 3;
    VIPMI_TRY(S.rayleigh_ritz(S.X[iy], S.X[iq], bact, nlock, th, res, &ok));
    qi = iq;
    if (!ok) { reason = 3; break; }
  }
  if (!reason && nlock < k) reason = 2;
  if (info) { info[0] = S.matvecs; info[1] = S.rounds; info[2] = nlock; info[3] = reason; }
  if (reason) return VIPMI_OK;
  // order (descending) and write out
  int perm[BMAX];
  for (int j = 0; j < k; ++j) perm[j] = j;
  std::stable_sort(perm, perm + k, [&](int a_, int b_) { return lth[a_] > lth[b_]; });
  VIPMI_CHECK_HIP(hipMemcpyAsync(perm_d, perm, sizeof(int) * k, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)k), dim3(256), 0, st, S.L, BMAX, n, perm_d, S.lth_d, evals, evecs);
  VIPMI_CHECK_HIP(hipGetLastError());
  VIPMI_CHECK_HIP(hipStreamSynchronize(st));            // (perm / lth live on this stack frame)
  *converged = 1;
  return VIPMI_OK;
}

}  // namespace vipmi
