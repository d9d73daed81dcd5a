// eigh_tri_large.hip -- leading k eigenpairs of ONE larger symmetric float64 matrix (128 <= n <= 6144, any k <= n;
// above 2048 rows: tri_xl_kernel, further down):
// the algorithm of eigh_tri.hip's multi-workgroup variant with the matrix left in global memory (a 2000 x 2000
// float64 matrix is 32 MB: it no longer fits the LDS of the cooperating workgroups, but it does fit the L2s and the
// Infinity Cache).  W = 64 workgroups own the rows cyclically (row r -> workgroup r mod W) and update them in place;
// per Householder step a workgroup reads and writes its share of the trailing matrix once (all loads of a row issued
// before the first use), then the same single counter barrier / redundant-reflector scheme as in eigh_tri.hip.
// Afterwards workgroup c computes eigenvalues c, c + W, ..., their eigenvectors by inverse iteration (factors in LDS)
// and their back-transformations; workgroup 0 finally orthonormalises the k vectors.
//
// Replaces, for C5-sized cubes (n = 2000 frames), the one-sided Jacobi kernel (140 ms) in the decomposition step of
// svd_wrapper / get_eigenvectors (psfsub/svd.py:342-702).
#include "common.h"
#include "wave_util.h"
#include "tri_common.h"

namespace vipmi {

namespace {

constexpr int LNT = 512;            // threads per workgroup (256-VGPR budget)
constexpr int LNW = LNT / 64;
constexpr double LEPS = tri::EPS;

template <int RPL>
__global__ __launch_bounds__(LNT) void tri_large_kernel(double* __restrict__ A, int n, int k, double* __restrict__ evals,
                                                        double* __restrict__ evecs, double* __restrict__ gb,
                                                        unsigned* __restrict__ bar, int all_evals, int* __restrict__ fail) {
  extern __shared__ double sm[];
  const int W = gridDim.x, wg = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // phase 1: six vectors of n ; later phases reuse the space (9 n doubles in total)
  double* vbuf0 = sm;
  double* vbuf1 = vbuf0 + n;
  double* vprev = vbuf1 + n;
  double* wprev = vprev + n;
  double* pfull = wprev + n;
  double* cfull = pfull + n;
  double* Pb = gb;                 // [2][n]
  double* Cb = gb + 2 * n;         // [2][n]
  double* Db = gb + 4 * n;         // [n]
  double* gd = gb + 5 * n;         // [n] diagonal of T       (written by workgroup 0)
  double* ge = gb + 6 * n;         // [n] off-diagonal
  double* gt = gb + 7 * n;         // [n] Householder scalars
  unsigned bar_target = 0;
  const int na = n;
  const int kk = k < na ? k : na;

  for (int c = tid; c < n; c += LNT) {
    cfull[c] = A[c];               // row 0: input data
    vprev[c] = 0.0;
    wprev[c] = 0.0;
    pfull[c] = 0.0;
  }
  __syncthreads();
  bar_target += W;
  grid_barrier(bar, bar_target, W, fail);            // everybody has row 0 before reflectors overwrite the matrix
  double* vcur = vbuf0;
  double* vnext = vbuf1;
  double beta_cur = 0.0;
  // (v, beta, alpha, diagonal) from the updated row held in cfull[c], c >= s1 ; returns beta (uniform)
  auto form_reflector = [&](int s1, double* vout) -> double {
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
    for (int c = s1 + 1 + tid; c < na; c += LNT) vout[c] = (c == s1 + 1) ? v0 : cfull[c];
    if (wg == 0 && tid == 0) {
      st_shared(&gd[s1], cfull[s1]);
      st_shared(&ge[s1], (nrm2 > 0.0) ? alpha : 0.0);
      st_shared(&gt[s1], beta);
    }
    return beta;
  };
  beta_cur = form_reflector(0, vcur);
  __syncthreads();

  // ---------------- 1. tridiagonalisation, matrix in global memory ----------------
  for (int s = 0; s + 2 < na; ++s) {
    const int par = s & 1;
    const double beta = beta_cur;
    if (s % W == wg)
      for (int c = s + 1 + tid; c < na; c += LNT) st_shared(&A[(size_t)s * n + c], vcur[c]);
    const int lr0 = (s + 1 - wg + W - 1) / W;
    for (int lr = (lr0 > 0 ? lr0 : 0) + wave;; lr += LNW) {
      const int r = lr * W + wg;
      if (r >= na) break;
      double* row = A + (size_t)r * n;
      double a[RPL];
#pragma unroll
      for (int ch = 0; ch < RPL; ++ch) {
        const int c = s + 1 + lane + 64 * ch;
        a[ch] = (c < na) ? row[c] : 0.0;
      }
      const double vr = vprev[r], wr = wprev[r];
      double acc = 0.0, cval = 0.0, dval = 0.0;
#pragma unroll
      for (int ch = 0; ch < RPL; ++ch) {
        const int c = s + 1 + lane + 64 * ch;
        if (c < na) {
          const double t = a[ch] - vr * wprev[c] - wr * vprev[c];
          row[c] = t;
          acc += t * vcur[c];
          if (ch == 0 && lane == 0) cval = t;
          if (c == r) dval = t;
        }
      }
      acc = wave_sum(acc);
      if (lane == 0) {
        st_shared(&Pb[par * n + r], beta * acc);
        st_shared(&Cb[par * n + r], cval);
      }
      if (s + 3 == na) {
        const int dl = r - s - 1;
        if (lane == (dl & 63)) st_shared(&Db[r], dval);
      }
    }
    bar_target += W;
    grid_barrier(bar, bar_target, W, fail);
    for (int c = s + 1 + tid; c < na; c += LNT) {
      pfull[c] = ld_shared(&Pb[par * n + c]);
      cfull[c] = ld_shared(&Cb[par * n + c]);
    }
    __syncthreads();
    double kd = 0.0;
    for (int r = s + 1 + lane; r < na; r += 64) kd += vcur[r] * pfull[r];
    const double K = 0.5 * beta * wave_sum(kd);
    const double vs1 = vcur[s + 1], ws1 = pfull[s + 1] - K * vs1;
    for (int r = s + 1 + tid; r < na; r += LNT) {
      const double v = vcur[r];
      const double w = pfull[r] - K * v;
      wprev[r] = w;
      vprev[r] = v;
      cfull[r] = cfull[r] - vs1 * w - ws1 * v;
    }
    __syncthreads();
    if (s + 3 < na) {
      beta_cur = form_reflector(s + 1, vnext);
      double* t = vcur;
      vcur = vnext;
      vnext = t;
    }
    __syncthreads();
  }
  if (wg == 0 && tid == 0) {
    const int a = na - 2, b = na - 1;
    st_shared(&gd[a], cfull[a]);
    st_shared(&ge[a], cfull[b]);
    st_shared(&gd[b], ld_shared(&Db[b]) - 2.0 * vprev[b] * wprev[b]);
    st_shared(&ge[b], 0.0);
  }
  bar_target += W;
  grid_barrier(bar, bar_target, W, fail);

  // ---------------- 2. T into LDS (scaled), eigenvalue of this workgroup's vector ----------------
  double* dd = sm;                 // [n]
  double* ee = dd + n;             // [n]
  double* e2 = ee + n;             // [n]
  double* lamv = e2 + n;           // [8]  (then the inverse-iteration arrays, 6 n)
  double* U0 = lamv + 8;
  double* U1 = U0 + n;
  double* U2 = U1 + n;
  double* Lm = U2 + n;
  double* Ls = Lm + n;
  double* Zl = Ls + n;
  for (int i = tid; i < na; i += LNT) {
    dd[i] = ld_shared(&gd[i]);
    ee[i] = ld_shared(&ge[i]);
  }
  __syncthreads();
  double scale = 0.0, glo = 0.0, ghi = 0.0;
  {
    double mx = 0.0;
    for (int i = lane; i < na; i += 64) mx = fmax(mx, fmax(fabs(dd[i]), fabs(ee[i])));
    scale = wave_max(mx);
  }
  const double iscale = scale > 0.0 ? 1.0 / scale : 0.0;
  __syncthreads();
  for (int i = tid; i < na; i += LNT) {
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
    const double margin = 4.0 * LEPS * (double)na + 1e-290;
    glo -= margin;
    ghi += margin;
  }
  // vectors c = wg, wg + W, ... of this workgroup, one after the other (k <= W: one each).  Phase 4 overwrites dd with
  // the Householder scalars, so every further vector starts by restoring the scaled diagonal.
  for (int c = wg, first = 1; first || c < kk; c += W, first = 0) {
  const bool mine = c < kk;
  if (!first) {
    __syncthreads();
    for (int i = tid; i < na; i += LNT) dd[i] = ld_shared(&gd[i]) * iscale;
    __syncthreads();
  }
  if (mine && wave == 0) {
    const int target = na - 1 - c;
    const double lam_ = tri::multisect(dd, e2, na, target, glo, ghi, lane);
    if (lane == 0) lamv[0] = lam_;
  }
  if (all_evals && first) {                // the rest of the spectrum (values only): waves 1.. of every workgroup
    for (int i = kk + wg * (LNW - 1) + (wave - 1); wave > 0 && i < na; i += W * (LNW - 1)) {
      const int target = na - 1 - i;
      const double lam_ = tri::multisect(dd, e2, na, target, glo, ghi, lane);
      if (lane == 0) evals[i] = lam_ * scale;
    }
  }
  __syncthreads();

  // ---------------- 3. inverse iteration (one thread; factors in LDS) ----------------
  if (mine && tid == 0) {
    const double lc = lamv[0] - (double)(c + 1) * 4.0 * LEPS;
    const double ptiny = 1e-3 * LEPS;
    double p = dd[0] - lc, q = (na > 1) ? ee[0] : 0.0, r = 0.0;
    double yc = tri::hash_unit(0u, (unsigned)c);
    for (int i = 0; i + 1 < na; ++i) {
      const double sub = ee[i], nd = dd[i + 1] - lc, nu = (i + 2 < na) ? ee[i + 1] : 0.0;
      const double yn = tri::hash_unit((unsigned)(i + 1), (unsigned)c);
      double inv, u1, u2, yi, m, sw;
      if (fabs(sub) > fabs(p) && fabs(sub) >= ptiny) {
        inv = tri::fast_rcp(sub);
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
        inv = tri::fast_rcp(p);
        u1 = q; u2 = r;
        m = sub * inv;
        sw = 0.0;
        yi = yc;
        yc = yn - m * yc;
        p = nd - m * q;
        q = nu - m * r;
        r = 0.0;
      }
      U0[i] = inv;
      U1[i] = u1;
      U2[i] = u2;
      Lm[i] = m;
      Ls[i] = sw;
      Zl[i] = yi;
    }
    if (fabs(p) < ptiny) p = (p < 0.0) ? -ptiny : ptiny;
    const double invlast = tri::fast_rcp(p);
    double rs = 1.0;
    for (int it = 0; it < 2; ++it) {
      if (it > 0) {
        yc = Zl[0] * rs;
        for (int i = 0; i + 1 < na; ++i) {
          const double yn = Zl[i + 1] * rs;
          const double m = Lm[i];
          const bool sw = Ls[i] != 0.0;
          const double yi = sw ? yn : yc;
          yc = sw ? (yc - m * yn) : (yn - m * yc);
          Zl[i] = yi;
        }
      }
      double x1 = yc * invlast, x2 = 0.0;
      Zl[na - 1] = x1;
      double acc = x1 * x1;
      for (int i = na - 2; i >= 0; --i) {
        const double x = (Zl[i] - U1[i] * x1 - U2[i] * x2) * U0[i];
        Zl[i] = x;
        acc += x * x;
        x2 = x1;
        x1 = x;
      }
      rs = acc > 0.0 ? 1.0 / sqrt(acc) : 1.0;
    }
    lamv[1] = rs;
  }
  __syncthreads();

  // ---------------- 4. back-transformation of this workgroup's vector: rows split over the 8 waves ----------------
  // z lives in LDS (Zl); every reflector needs one dot product over the whole vector: waves reduce their slices,
  // partial sums through LDS.  Reflectors are staged in blocks of RB rows.
  {
    constexpr int RB = 4;
    double* stage = U0;                         // [RB][n] over U0..Lm (dead)
    double* part = Zl + n;                      // [2][LNW] partial dots
    if (mine) {
      const double rs = lamv[1];
      for (int i = tid; i < na; i += LNT) {
        Zl[i] *= rs;
        dd[i] = ld_shared(&gt[i]);              // Householder scalars (dd is dead)
      }
      for (int jb = na - 3; jb >= 0; jb -= RB) {
        __syncthreads();
        for (int e = tid; e < RB * n; e += LNT) {
          const int q = e / n, i = e - q * n, j = jb - q;
          stage[e] = (j >= 0 && i > j && i < na) ? ld_shared(&A[(size_t)j * n + i]) : 0.0;
        }
        __syncthreads();
        for (int q = 0; q < RB; ++q) {
          const int j = jb - q;
          if (j < 0) break;
          const double beta = dd[j];
          // every thread owns the same elements i = tid + LNT m in all steps: no cross-thread hazard on Zl
          double sdot = 0.0;
          for (int i = tid; i < na; i += LNT) sdot += stage[q * n + i] * Zl[i];      // stage is 0 for i <= j
          sdot = wave_sum(sdot);
          if (lane == 0) part[wave + (q & 1) * LNW] = sdot;
          __syncthreads();
          double tot = 0.0;
#pragma unroll
          for (int w = 0; w < LNW; ++w) tot += part[w + (q & 1) * LNW];
          tot *= beta;
          for (int i = tid; i < na; i += LNT) Zl[i] -= tot * stage[q * n + i];
        }
      }
    }
    __syncthreads();
    if (mine) {
      for (int i = tid; i < n; i += LNT) st_shared(&evecs[(size_t)c * n + i], (i < na) ? Zl[i] : 0.0);
      if (tid == 0) st_shared(&evals[c], lamv[0] * scale);
    }
  }
  }   // vectors of this workgroup
  bar_target += W;
  grid_barrier(bar, bar_target, W, fail);
  if (wg != 0) return;

  // ---------------- 5. workgroup 0: modified Gram-Schmidt in place (vectors in global / L2), sign convention ---------
  double* qv = sm;                               // [n] pivot vector
  for (int cp = 0; cp < kk; ++cp) {
    // normalise vector cp (all threads)
    double sq = 0.0;
    for (int i = tid; i < na; i += LNT) {
      const double x = ld_shared(&evecs[(size_t)cp * n + i]);
      qv[i] = x;
      sq += x * x;
    }
    sq = wave_sum(sq);
    __syncthreads();                             // qv complete ; reuse part area for the block reduction
    double* red = sm + n;
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < LNW; ++w) tot += red[w];
    const double inv = tot > 0.0 ? 1.0 / sqrt(tot) : 0.0;
    __syncthreads();
    for (int i = tid; i < na; i += LNT) {
      const double x = qv[i] * inv;
      qv[i] = x;
      st_shared(&evecs[(size_t)cp * n + i], x);
    }
    __syncthreads();
    // remove its component from the later vectors: one wave per vector
    for (int c2 = cp + 1 + wave; c2 < kk; c2 += LNW) {
      double x[RPL], sdot = 0.0;
#pragma unroll
      for (int rr = 0; rr < RPL; ++rr) {
        const int i = lane + 64 * rr;
        x[rr] = (i < na) ? ld_shared(&evecs[(size_t)c2 * n + i]) : 0.0;
        sdot += (i < na) ? x[rr] * qv[i] : 0.0;
      }
      sdot = wave_sum(sdot);
#pragma unroll
      for (int rr = 0; rr < RPL; ++rr) {
        const int i = lane + 64 * rr;
        if (i < na) st_shared(&evecs[(size_t)c2 * n + i], x[rr] - sdot * qv[i]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // sign convention: largest-magnitude component positive ; one wave per vector
  for (int c2 = wave; c2 < k; c2 += LNW) {
    double x[RPL];
    double best = -1.0, bval = 0.0;
    int bidx = 0x7fffffff;
#pragma unroll
    for (int rr = 0; rr < RPL; ++rr) {
      const int i = lane + 64 * rr;
      x[rr] = (c2 < kk && i < na) ? ld_shared(&evecs[(size_t)c2 * n + i]) : 0.0;
      const double a = fabs(x[rr]);
      if (i < na && a > best) {
        best = a;
        bval = x[rr];
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
    const double sg = (c2 < kk) ? (bval < 0.0 ? -1.0 : 1.0) : 0.0;
#pragma unroll
    for (int rr = 0; rr < RPL; ++rr) {
      const int i = lane + 64 * rr;
      if (i < n) evecs[(size_t)c2 * n + i] = (i < na) ? x[rr] * sg : 0.0;
    }
    if (lane == 0 && c2 >= kk) evals[c2] = 0.0;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// More than 2048 rows (up to 512 * 12 = 6144 with the vectors in LDS, 16384 with them in global memory): the same algorithm with THREE vectors of n doubles in LDS instead of
// nine.  Tridiagonalisation: the gathered vectors p and c (column s+1) stay in registers (element c = s+1 + tid + 512 j
// belongs to thread tid: 12 per thread at most), the reductions over them are workgroup sums, and the next reflector is
// written into the buffer of v_{s-1}, which is dead once the row pass of the step is over.  Later phases: d, e and the
// vector under construction in LDS; the Sturm counts square e as they fetch it; the factors of the inverse iteration
// live in global memory (written once, read back in blocks of 16 rows); the back-transformation holds the vector in
// registers and reads the reflectors from the (L2 / Infinity-Cache resident) matrix, the next one prefetched.
// XPT: elements of a gathered vector per thread (n <= 512 XPT).  GV (round 6, more than 6144 rows): the three n-vectors of the
// workgroup live in GLOBAL memory (gvec: 3 n doubles per workgroup, L2-resident) instead of LDS -- the slow, correct path for any
// matrix that fits the device: every vector element of the row pass is one more load, the reductions are unchanged.
constexpr int XL_XPT_MAX = 32;          // 512 threads x 32 elements: matrices of up to 16384 rows (2.1 GB)
constexpr int XBS = 16;                 // rows per block of the substitutions
constexpr int XCG = 16;                 // chunks of 64 columns per group of the row pass

// deterministic workgroup sum through 8 LDS slots; `slot` alternates so that one barrier per sum is enough
__device__ __forceinline__ double xl_block_sum(double x, double* __restrict__ red, int slot, int lane, int wave) {
  x = wave_sum(x);
  if (lane == 0) red[slot * LNW + wave] = x;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < LNW; ++w) t += red[slot * LNW + w];
  return t;
}

template <int XPT, bool GV>
__global__ __launch_bounds__(LNT) void tri_xl_kernel(double* __restrict__ A, int n, int k, double* __restrict__ evals,
                                                     double* __restrict__ evecs, double* __restrict__ gb,
                                                     double* __restrict__ scr_all, unsigned* __restrict__ bar,
                                                     int all_evals, int* __restrict__ fail, double* __restrict__ gvec) {
  extern __shared__ double sm[];
  const int W = gridDim.x, wg = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* const vec3 = GV ? gvec + (size_t)wg * 3 * n : sm;      // the workgroup's three n-vectors
  double* vb0 = vec3;              // v_s / v_{s-1}, rotating
  double* vb1 = vb0 + n;
  double* wprev = vb1 + n;
  double* red = GV ? sm : wprev + n;         // [6][LNW] reduction slots, then [8] broadcast values (always LDS)
  double* bc = red + 6 * LNW;
  double* Pb = gb;                 // [2][n]
  double* Cb = gb + 2 * n;         // [2][n]
  double* Db = gb + 4 * n;         // [n]
  double* gd = gb + 5 * n;         // [n] diagonal of T       (written by workgroup 0)
  double* ge = gb + 6 * n;         // [n] off-diagonal
  double* gt = gb + 7 * n;         // [n] Householder scalars
  unsigned bar_target = 0;
  const int na = n;
  const int kk = k < na ? k : na;

  for (int c = tid; c < n; c += LNT) {
    vb0[c] = 0.0;
    vb1[c] = 0.0;
    wprev[c] = 0.0;
  }
  double cj[XPT], pj[XPT];
  // (ownership base b: thread tid holds elements c = b + tid + 512 j; b = s + 1 in step s, 0 before the first step)
#pragma unroll
  for (int j = 0; j < XPT; ++j) {
    const int c = tid + LNT * j;
    cj[j] = (c < na) ? A[c] : 0.0;                 // row 0: input data
    pj[j] = 0.0;
  }
  __syncthreads();
  bar_target += W;
  grid_barrier(bar, bar_target, W, fail);                // everybody has row 0 before reflectors overwrite the matrix
  double* vprev = vb0;                             // v_{s-1} (zeros before the first step)
  double* vcur = vb1;
  int rs = 0;                                      // reduction slot, cycles 0..5
  // reflector from the column held in cj with ownership base b (= its diagonal index): v into vout[c > b], returns beta
  auto form_reflector = [&](int b, double* vout) -> double {
    double nrm2 = 0.0;
#pragma unroll
    for (int j = 0; j < XPT; ++j) {
      const int c = b + tid + LNT * j;
      if (c > b && c < na) nrm2 += cj[j] * cj[j];
    }
    if (tid == 1) bc[0] = cj[0];                   // x0 = element b + 1
    nrm2 = xl_block_sum(nrm2, red, rs, lane, wave);
    rs = (rs + 1) % 6;
    const double x0 = bc[0];
    const double nrm = sqrt(nrm2);
    const double alpha = (x0 >= 0.0) ? -nrm : nrm;
    const double v0 = x0 - alpha;
    double rest = nrm2 - x0 * x0;
    if (rest < 0.0) rest = 0.0;
    const double vv = rest + v0 * v0;
    const double beta = (nrm2 > 0.0 && vv > 0.0) ? 2.0 / vv : 0.0;
#pragma unroll
    for (int j = 0; j < XPT; ++j) {
      const int c = b + tid + LNT * j;
      if (c > b && c < na) vout[c] = (c == b + 1) ? v0 : cj[j];
    }
    if (wg == 0 && tid == 0) {
      st_shared(&gd[b], cj[0]);
      st_shared(&ge[b], (nrm2 > 0.0) ? alpha : 0.0);
      st_shared(&gt[b], beta);
    }
    return beta;
  };
  double beta_cur = form_reflector(0, vcur);
  __syncthreads();

  // ---------------- 1. tridiagonalisation, matrix in global memory ----------------
  for (int s = 0; s + 2 < na; ++s) {
    const int par = s & 1;
    const double beta = beta_cur;
    if (s % W == wg)
      for (int c = s + 1 + tid; c < na; c += LNT) st_shared(&A[(size_t)s * n + c], vcur[c]);
    const int lr0 = (s + 1 - wg + W - 1) / W;
    for (int lr = (lr0 > 0 ? lr0 : 0) + wave;; lr += LNW) {
      const int r = lr * W + wg;
      if (r >= na) break;
      double* row = A + (size_t)r * n;
      const double vr = vprev[r], wr = wprev[r];
      double acc = 0.0, cval = 0.0, dval = 0.0;
      // the row in groups of XCG chunks of 64 columns, the next group in flight while this one is updated
      double a[XCG], an[XCG];
#pragma unroll
      for (int ch = 0; ch < XCG; ++ch) {
        const int c = s + 1 + lane + 64 * ch;
        a[ch] = (c < na) ? row[c] : 0.0;
      }
      for (int c0 = s + 1; c0 < na; c0 += 64 * XCG) {
#pragma unroll
        for (int ch = 0; ch < XCG; ++ch) {
          const int c = c0 + 64 * XCG + lane + 64 * ch;
          an[ch] = (c < na) ? row[c] : 0.0;
        }
#pragma unroll
        for (int ch = 0; ch < XCG; ++ch) {
          const int c = c0 + lane + 64 * ch;
          if (c < na) {
            const double t = a[ch] - vr * wprev[c] - wr * vprev[c];
            row[c] = t;
            acc += t * vcur[c];
            if (c == s + 1) cval = t;
            if (c == r) dval = t;
          }
          a[ch] = an[ch];
        }
      }
      acc = wave_sum(acc);
      if (lane == 0) {                             // (column s + 1 is lane 0 of the first chunk)
        st_shared(&Pb[par * n + r], beta * acc);
        st_shared(&Cb[par * n + r], cval);
      }
      if (s + 3 == na) {
        const int dl = r - s - 1;
        if (lane == (dl & 63)) st_shared(&Db[r], dval);
      }
    }
    bar_target += W;
    grid_barrier(bar, bar_target, W, fail);
    double kd = 0.0;
#pragma unroll
    for (int j = 0; j < XPT; ++j) {
      const int c = s + 1 + tid + LNT * j;
      if (c < na) {
        pj[j] = ld_shared(&Pb[par * n + c]);
        cj[j] = ld_shared(&Cb[par * n + c]);
        kd += vcur[c] * pj[j];
      } else {
        pj[j] = 0.0;
        cj[j] = 0.0;
      }
    }
    if (tid == 0) bc[1] = pj[0];                   // p[s + 1]
    const double K = 0.5 * beta * xl_block_sum(kd, red, rs, lane, wave);
    rs = (rs + 1) % 6;
    const double vs1 = vcur[s + 1], ws1 = bc[1] - K * vs1;
#pragma unroll
    for (int j = 0; j < XPT; ++j) {
      const int c = s + 1 + tid + LNT * j;
      if (c < na) {
        const double v = vcur[c];
        const double w = pj[j] - K * v;
        wprev[c] = w;
        cj[j] = cj[j] - vs1 * w - ws1 * v;
      }
    }
    // v_s becomes the pending vector; the next reflector goes where v_{s-1} was (its last readers passed the barrier)
    double* vfree = vprev;
    vprev = vcur;
    if (s + 3 < na) {
      beta_cur = form_reflector(s + 1, vfree);
      vcur = vfree;
    }
    __syncthreads();
  }
  if (wg == 0) {
    // after the last step (s = na - 3, base na - 2): thread 0 holds element a = na - 2, thread 1 element b = na - 1
    const int a = na - 2, b = na - 1;
    if (na >= 3) {
      if (tid == 0) st_shared(&gd[a], cj[0]);
      if (tid == 1) {
        st_shared(&ge[a], cj[0]);
        st_shared(&gd[b], ld_shared(&Db[b]) - 2.0 * vprev[b] * wprev[b]);
        st_shared(&ge[b], 0.0);
      }
    }
  }
  bar_target += W;
  grid_barrier(bar, bar_target, W, fail);

  // ---------------- 2. T into LDS (scaled) ----------------
  double* dd = vec3;               // [n]
  double* ee = dd + n;             // [n]
  double* Zl = ee + n;             // [n] right-hand side / solution of the inverse iteration
  double* lamv = bc;               // [8]
  double* scr = scr_all + (size_t)wg * 5 * n;
  double* __restrict__ U0 = scr;           // reciprocal pivots
  double* __restrict__ U1 = scr + n;       // first superdiagonal of U
  double* __restrict__ U2 = scr + 2 * n;   // second superdiagonal (row swaps)
  double* __restrict__ Lm = scr + 3 * n;   // multipliers of L
  double* __restrict__ Ls = scr + 4 * n;   // 1 where rows i, i+1 were swapped
  __syncthreads();
  for (int i = tid; i < na; i += LNT) {
    dd[i] = ld_shared(&gd[i]);
    ee[i] = ld_shared(&ge[i]);
  }
  __syncthreads();
  double scale = 0.0, glo = 0.0, ghi = 0.0;
  {
    double mx = 0.0;
    for (int i = lane; i < na; i += 64) mx = fmax(mx, fmax(fabs(dd[i]), fabs(ee[i])));
    scale = wave_max(mx);
  }
  const double iscale = scale > 0.0 ? 1.0 / scale : 0.0;
  __syncthreads();
  for (int i = tid; i < na; i += LNT) {
    dd[i] *= iscale;
    ee[i] *= iscale;
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
    const double margin = 4.0 * LEPS * (double)na + 1e-290;
    glo -= margin;
    ghi += margin;
  }
  // vectors c = wg, wg + W, ... of this workgroup, one after the other
  for (int c = wg, first = 1; first || c < kk; c += W, first = 0) {
    const bool mine = c < kk;
    if (!first) {
      __syncthreads();
      for (int i = tid; i < na; i += LNT) dd[i] = ld_shared(&gd[i]) * iscale;   // (phase 4 keeps the reflector scalars in dd)
      __syncthreads();
    }
    if (mine && wave == 0) {
      const double lam_ = tri::multisect<true>(dd, ee, na, na - 1 - c, glo, ghi, lane);
      if (lane == 0) lamv[0] = lam_;
    }
    if (all_evals && first) {              // the rest of the spectrum (values only): waves 1.. of every workgroup
      for (int i = kk + wg * (LNW - 1) + (wave - 1); wave > 0 && i < na; i += W * (LNW - 1)) {
        const double lam_ = tri::multisect<true>(dd, ee, na, na - 1 - i, glo, ghi, lane);
        if (lane == 0) evals[i] = lam_ * scale;
      }
    }
    __syncthreads();

    // ---------------- 3. inverse iteration (one thread; factors in global memory, vector in LDS) ----------------
    if (mine && tid == 0) {
      const double lc = lamv[0] - (double)(c + 1) * 4.0 * LEPS;
      const double ptiny = 1e-3 * LEPS;
      double p = dd[0] - lc, q = (na > 1) ? ee[0] : 0.0, r = 0.0;
      double yc = tri::hash_unit(0u, (unsigned)c);
      for (int i = 0; i + 1 < na; ++i) {
        const double sub = ee[i], nd = dd[i + 1] - lc, nu = (i + 2 < na) ? ee[i + 1] : 0.0;
        const double yn = tri::hash_unit((unsigned)(i + 1), (unsigned)c);
        double inv, u1, u2, yi, m, sw;
        if (fabs(sub) > fabs(p) && fabs(sub) >= ptiny) {
          inv = tri::fast_rcp(sub);
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
          inv = tri::fast_rcp(p);
          u1 = q; u2 = r;
          m = sub * inv;
          sw = 0.0;
          yi = yc;
          yc = yn - m * yc;
          p = nd - m * q;
          q = nu - m * r;
          r = 0.0;
        }
        U0[i] = inv;
        U1[i] = u1;
        U2[i] = u2;
        Lm[i] = m;
        Ls[i] = sw;
        Zl[i] = yi;
      }
      if (fabs(p) < ptiny) p = (p < 0.0) ? -ptiny : ptiny;
      const double invlast = tri::fast_rcp(p);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the factors are read back by this thread
      double rs_ = 1.0;
      for (int it = 0; it < 2; ++it) {
        if (it > 0) {
          // forward substitution of the previous solution (scaled to unit norm) through P L, rows in blocks of XBS
          yc = Zl[0] * rs_;
          for (int i0 = 0; i0 + 1 < na; i0 += XBS) {
            double mb[XBS], sb[XBS];
#pragma unroll
            for (int u = 0; u < XBS; ++u) {
              const int i = i0 + u;
              mb[u] = (i + 1 < na) ? Lm[i] : 0.0;
              sb[u] = (i + 1 < na) ? Ls[i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < XBS; ++u) {
              const int i = i0 + u;
              if (i + 1 < na) {
                const double yn = Zl[i + 1] * rs_;
                const bool sw = sb[u] != 0.0;
                const double yi = sw ? yn : yc;
                yc = sw ? (yc - mb[u] * yn) : (yn - mb[u] * yc);
                Zl[i] = yi;
              }
            }
          }
        }
        // back substitution, rows na-2 .. 0 in blocks of XBS
        double x1 = yc * invlast, x2 = 0.0;
        Zl[na - 1] = x1;
        double acc = x1 * x1;
        for (int i0 = na - 2; i0 >= 0; i0 -= XBS) {
          double a0[XBS], a1[XBS], a2[XBS];
#pragma unroll
          for (int u = 0; u < XBS; ++u) {
            const int i = i0 - u;
            a0[u] = (i >= 0) ? U0[i] : 0.0;
            a1[u] = (i >= 0) ? U1[i] : 0.0;
            a2[u] = (i >= 0) ? U2[i] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < XBS; ++u) {
            const int i = i0 - u;
            if (i >= 0) {
              const double x = (Zl[i] - a1[u] * x1 - a2[u] * x2) * a0[u];
              Zl[i] = x;
              acc += x * x;
              x2 = x1;
              x1 = x;
            }
          }
        }
        rs_ = acc > 0.0 ? 1.0 / sqrt(acc) : 1.0;
      }
      lamv[1] = rs_;
    }
    __syncthreads();

    // ---------------- 4. back-transformation: the vector in registers (element i = tid + 512 m of thread tid in every
    // step: no cross-thread hazard), reflectors straight from the matrix, the next one in flight ----------------
    if (mine) {
      const double rsn = lamv[1];
      double z[XPT], vj[XPT];
#pragma unroll
      for (int m = 0; m < XPT; ++m) {
        const int i = tid + LNT * m;
        z[m] = (i < na) ? Zl[i] * rsn : 0.0;
      }
      __syncthreads();
      for (int i = tid; i < na; i += LNT) dd[i] = ld_shared(&gt[i]);      // Householder scalars (dd is restored per vector)
      int jn = na - 3;
#pragma unroll
      for (int m = 0; m < XPT; ++m) {
        const int i = tid + LNT * m;
        vj[m] = (jn >= 0 && i > jn && i < na) ? ld_shared(&A[(size_t)jn * n + i]) : 0.0;
      }
      __syncthreads();
      for (int j = na - 3; j >= 0; --j) {
        double vn[XPT];
#pragma unroll
        for (int m = 0; m < XPT; ++m) {
          const int i = tid + LNT * m;
          vn[m] = (j >= 1 && i > j - 1 && i < na) ? ld_shared(&A[(size_t)(j - 1) * n + i]) : 0.0;
        }
        double sdot = 0.0;
#pragma unroll
        for (int m = 0; m < XPT; ++m) sdot += vj[m] * z[m];
        const double tot = dd[j] * xl_block_sum(sdot, red, j & 1, lane, wave);
#pragma unroll
        for (int m = 0; m < XPT; ++m) {
          z[m] -= tot * vj[m];
          vj[m] = vn[m];
        }
      }
      for (int m = 0; m < XPT; ++m) {
        const int i = tid + LNT * m;
        if (i < n) st_shared(&evecs[(size_t)c * n + i], (i < na) ? z[m] : 0.0);
      }
      if (tid == 0) st_shared(&evals[c], lamv[0] * scale);
    }
  }   // vectors of this workgroup
  bar_target += W;
  grid_barrier(bar, bar_target, W, fail);
  if (wg != 0) return;

  // ---------------- 5. workgroup 0: modified Gram-Schmidt in place (vectors in global / L2), sign convention ---------
  double* qv = vec3;                             // [n] pivot vector
  double* red5 = GV ? sm : sm + n;
  for (int cp = 0; cp < kk; ++cp) {
    double sq = 0.0;
    for (int i = tid; i < na; i += LNT) {
      const double x = ld_shared(&evecs[(size_t)cp * n + i]);
      qv[i] = x;
      sq += x * x;
    }
    const double tot = xl_block_sum(sq, red5, cp & 1, lane, wave);
    const double inv = tot > 0.0 ? 1.0 / sqrt(tot) : 0.0;
    for (int i = tid; i < na; i += LNT) {      // (own elements only: written and read by the same thread)
      const double x = qv[i] * inv;
      qv[i] = x;
      st_shared(&evecs[(size_t)cp * n + i], x);
    }
    __syncthreads();
    // remove its component from the later vectors: one wave per vector, two passes over the vector
    for (int c2 = cp + 1 + wave; c2 < kk; c2 += LNW) {
      double sdot = 0.0;
      for (int i = lane; i < na; i += 64) sdot += ld_shared(&evecs[(size_t)c2 * n + i]) * qv[i];
      sdot = wave_sum(sdot);
      for (int i = lane; i < na; i += 64) {
        const double x = ld_shared(&evecs[(size_t)c2 * n + i]);
        st_shared(&evecs[(size_t)c2 * n + i], x - sdot * qv[i]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // sign convention: largest-magnitude component positive (lowest index on ties); one wave per vector
  for (int c2 = wave; c2 < k; c2 += LNW) {
    double best = -1.0, bval = 0.0;
    int bidx = 0x7fffffff;
    if (c2 < kk) {
      for (int i = lane; i < na; i += 64) {
        const double x = ld_shared(&evecs[(size_t)c2 * n + i]);
        const double a = fabs(x);
        if (a > best) {
          best = a;
          bval = x;
          bidx = i;
        }
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
    const double sg = (c2 < kk) ? (bval < 0.0 ? -1.0 : 1.0) : 0.0;
    for (int i = lane; i < n; i += 64) {
      const double x = (c2 < kk && i < na) ? ld_shared(&evecs[(size_t)c2 * n + i]) : 0.0;
      evecs[(size_t)c2 * n + i] = x * sg;
    }
    if (lane == 0 && c2 >= kk) evals[c2] = 0.0;
  }
}

int launch_xl(vipmi_ctx* ctx, double* A, int n, int k, double* evals, double* evecs, int all_evals) {
  int W = 128;
  if (W > ctx->num_cu) W = 64;
  double *gbuf = nullptr, *scr = nullptr, *gvec = nullptr;
  unsigned* bars = nullptr;
  const bool gv = n > 512 * 12;                        // beyond 6144 rows the three n-vectors no longer fit the LDS
  VIPMI_TRY(ws(ctx, "eigh_large_gbuf", (size_t)8 * n, &gbuf));
  VIPMI_TRY(ws(ctx, "eigh_xl_factors", (size_t)W * 5 * n, &scr));
  VIPMI_TRY(ws(ctx, "eigh_large_bar", (size_t)1, &bars));
  if (gv) VIPMI_TRY(ws(ctx, "eigh_xl_vectors", (size_t)W * 3 * n, &gvec));
  VIPMI_CHECK_HIP(hipMemsetAsync(bars, 0, sizeof(unsigned), ctx->stream));
  int* fail = nullptr;                                  // barrier time-outs are latched here (vipmi_check_deferred)
  VIPMI_TRY(deferred_fail_words(ctx, &fail));
  const size_t lds = ((size_t)(gv ? 0 : 3 * n) + 6 * LNW + 8 + 16) * sizeof(double);
  VIPMI_REQUIRE(lds <= 160 * 1024, "eigh_topk(xl): LDS budget exceeded (%zu)", lds);
  auto kern = gv ? tri_xl_kernel<XL_XPT_MAX, true> : tri_xl_kernel<12, false>;
  VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds));
  hipLaunchKernelGGL(kern, dim3(W), dim3(LNT), lds, ctx->stream, A, n, k, evals, evecs, gbuf, scr, bars, all_evals, fail, gvec);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

template <int RPL>
int launch_large(vipmi_ctx* ctx, double* A, int n, int k, double* evals, double* evecs, int all_evals) {
  // workgroups (all co-resident: one per CU): a step is bound by the L2 bandwidth of the CUs that take part
  int W = (int)ctx->opt("eigh_large_w", 0);
  if (W != 32 && W != 64 && W != 128 && W != 256) W = n > 1024 ? 128 : 64;    // n = 2000: 43.0 / 35.7 / 35.1 ms with 64 / 128 / 256,
  if (W > ctx->num_cu) W = 64;                                                 // n = 1000: 10.6 / 10.1 / 11.2 (barrier-bound)
  double* gbuf = nullptr;
  unsigned* bars = nullptr;
  VIPMI_TRY(ws(ctx, "eigh_large_gbuf", (size_t)8 * n, &gbuf));
  VIPMI_TRY(ws(ctx, "eigh_large_bar", (size_t)1, &bars));
  VIPMI_CHECK_HIP(hipMemsetAsync(bars, 0, sizeof(unsigned), ctx->stream));
  int* fail = nullptr;                                  // barrier time-outs are latched here (vipmi_check_deferred)
  VIPMI_TRY(deferred_fail_words(ctx, &fail));
  const size_t lds = ((size_t)9 * n + 8 + 2 * LNW + 16) * sizeof(double);
  VIPMI_REQUIRE(lds <= 160 * 1024, "eigh_topk(large): LDS budget exceeded (%zu)", lds);
  auto kern = tri_large_kernel<RPL>;
  VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds));
  hipLaunchKernelGGL(kern, dim3(W), dim3(LNT), lds, ctx->stream, A, n, k, evals, evecs, gbuf, bars, all_evals, fail);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

}  // namespace

// (n below 512 is served when more than 64 vectors are wanted: the LDS-resident solvers of eigh_tri.hip stop there)
bool eigh_large_supported(int64_t n, int64_t k) { return n >= 128 && n <= 512 * XL_XPT_MAX && k >= 1 && k <= n; }

// one problem (batch entries are solved one after the other)
int eigh_large_f64(vipmi_ctx* ctx, double* A, int64_t batch, int64_t n, int64_t k, double* evals, double* evecs,
                   bool all_evals) {
  VIPMI_REQUIRE(A && evals && evecs, "eigh_large: null pointer");
  VIPMI_REQUIRE(batch > 0 && eigh_large_supported(n, k), "eigh_large: unsupported sizes n=%ld k=%ld", (long)n, (long)k);
  StageScope sc(ctx, "eigh");
  for (int64_t p = 0; p < batch; ++p) {
    double* Ap = A + (size_t)p * n * n;
    double* ev = evals + (size_t)p * n;
    double* ec = evecs + (size_t)p * n * n;
    if (n > ctx->opt("eigh_xl_min", 900)) {      // (also the faster one from ~1000 rows: 30.7 against 35.0 ms at n = 2000, k = 50)
      VIPMI_TRY(launch_xl(ctx, Ap, (int)n, (int)k, ev, ec, all_evals));
    } else if (n <= 1024) {
      VIPMI_TRY(launch_large<16>(ctx, Ap, (int)n, (int)k, ev, ec, all_evals));
    } else {
      VIPMI_TRY(launch_large<32>(ctx, Ap, (int)n, (int)k, ev, ec, all_evals));
    }
  }
  return VIPMI_OK;
}

}  // namespace vipmi
