// eigh.hip -- batched symmetric eigendecomposition of small (n <= ~1280) float64 matrices by a
// one-sided block Jacobi method, written for CDNA4 (wave64, 160 KB LDS, agent-scope barriers).
//
// Replaces the LAPACK calls of svd_wrapper: `linalg.eigh(C)` (psfsub/svd.py:450, mode='eigen')
// and, via G = M M^T, the thin SVD `linalg.svd(matrix.T)` (svd.py:470, mode='lapack'): the
// eigenvectors of G are the left singular vectors of M and the PCs are S^-1 E^T M.
//
// Method.  One-sided (Hestenes) Jacobi on the columns of A = G: plane rotations make the columns
// of A V mutually orthogonal; then (A V)_j = lambda_j v_j, so lambda_j = |a_j| and v_j = a_j/|a_j|.
// No V accumulation is needed (G is symmetric positive semi-definite).
//   * columns are grouped in blocks of B = 16; a round-robin tournament pairs the blocks; one
//     workgroup (B waves) handles one block pair per round.  Wave w keeps column w of block I in
//     registers for the whole round and meets every column of block J (staged in LDS) once:
//     B steps of B disjoint rotations, one __syncthreads() per step.  The three dot products of a
//     rotation are wave reductions; nothing leaves the CU inside a round.
//   * pairs inside a block are swept once per sweep from LDS (home blocks of each workgroup).
//   * rounds are separated by an agent-scope release/acquire counter barrier (all workgroups of a
//     problem are co-resident: <= 40 workgroups of <= 160 KB LDS; the host launches chunks that fit).
//   * convergence: max |a_p.a_q| / (|a_p||a_q|) over a sweep < tol, checked uniformly after the
//     sweep's last barrier.
#include "common.h"
#include "wave_util.h"

namespace vipmi {

namespace {

// rotate the pair (ap, aq) held as RPL register rows per lane; returns |gamma|/sqrt(alpha beta).
// The rotation ANGLE only steers convergence, so tan(theta) is evaluated in float32 (v_rcp/v_sqrt,
// ~1e-7 relative); orthogonality is what accuracy needs, so c = 1/sqrt(1+t^2) is refined to float64 with
// two Newton steps from the float32 seed and s = c*t.  (IEEE float64 sqrt/div expansions used to be
// ~60 % of a Jacobi step.)
// null2: columns whose squared norm is below it are numerically null with respect to the MATRIX (rounding garbage of a rank-deficient
// Gram matrix): a pair of two such columns is left alone -- their inner products are noise that no rotation brings under a relative
// tolerance (300 sweeps did not, round 6), and every consumer drops eigenvalues below 1e-12 of the largest anyway.
template <int RPL>
__device__ __forceinline__ float rotate_pair(double (&ap)[RPL], double (&aq)[RPL], float tol, double null2) {
  double al = 0, be = 0, ga = 0;
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    al += ap[r] * ap[r];
    be += aq[r] * aq[r];
    ga += ap[r] * aq[r];
  }
  al = wave_sum(al);
  be = wave_sum(be);
  ga = wave_sum(ga);
  if (!(al > 0.0) || !(be > 0.0)) return 0.f;
  if (al < null2 && be < null2) return 0.f;
  // scale-free float32 quantities: r = beta/alpha, g = gamma/alpha (exponents removed in float64)
  const int ea = ilogb(al > be ? al : be);
  const float alf = (float)scalbn(al, -ea), bef = (float)scalbn(be, -ea), gaf = (float)scalbn(ga, -ea);
  if (!(alf > 1e-30f) || !(bef > 1e-30f)) return 0.f;   // one column is numerically null w.r.t. the other
  const float rel = fabsf(gaf) * __frsqrt_rn(alf) * __frsqrt_rn(bef);
  if (!(rel > tol)) return rel;
  // beta - alpha in float64 BEFORE the conversion: for eigenvalues that agree to more than seven digits (a cluster, duplicated
  // frames) the float32 values are equal or differ by rounding noise, the angle is arbitrary and the pair never converges
  // (clusters 5 +- 1e-9: stuck at 2e-10 for 300 sweeps, round 6)
  const float zeta = (float)scalbn(be - al, -ea) / (2.0f * gaf);
  float t = 1.0f / (fabsf(zeta) + sqrtf(1.0f + zeta * zeta));
  if (!(fabsf(zeta) < 1e18f)) t = 0.5f / fabsf(zeta);          // huge zeta: avoid inf*0
  if (zeta < 0.f) t = -t;
  const double td = (double)t;
  const double x = 1.0 + td * td;
  double c = (double)__frsqrt_rn((float)x);
  c = c * (1.5 - 0.5 * x * c * c);
  c = c * (1.5 - 0.5 * x * c * c);
  const double s = c * td;
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    const double xx = ap[r], yy = aq[r];
    ap[r] = c * xx - s * yy;
    aq[r] = s * xx + c * yy;
  }
  return rel;
}

__device__ __forceinline__ unsigned f2ord(float f) { return __float_as_uint(f); }  // f >= 0

template <int B, int RPL>
__global__ __launch_bounds__(64 * B) void jacobi_kernel(
    double* __restrict__ Gall, int n, int nblk, int max_sweeps, double tol, unsigned* __restrict__ bars,
    unsigned* __restrict__ conv, int* __restrict__ info, double* __restrict__ evals_all,
    double* __restrict__ evecs_all, double* __restrict__ norms_all, int prob0,
    int* __restrict__ fail, const double* __restrict__ null2_all) {
  extern __shared__ __attribute__((aligned(16))) double lds[];   // B columns x ldn
  const int prob = prob0 + blockIdx.y;
  const double null2 = null2_all[prob];
  const int g = blockIdx.x, nwg = gridDim.x;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* A = Gall + (size_t)prob * n * n;   // column j = A + j*n (G symmetric)
  unsigned* bar = bars + prob;
  unsigned* cv = conv + (size_t)prob * max_sweeps;
  const int ldn = RPL * 64;
  unsigned phase = 0;
  __shared__ float wg_max[B];

  auto load_col = [&](int col, double (&v)[RPL]) {
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      int i = lane + 64 * r;
      v[r] = (col < n && i < n) ? ld_shared(&A[(size_t)col * n + i]) : 0.0;
    }
  };
  auto store_col = [&](int col, const double (&v)[RPL]) {
    if (col >= n) return;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      int i = lane + 64 * r;
      if (i < n) st_shared(&A[(size_t)col * n + i], v[r]);
    }
  };
  auto lds_put = [&](int slot, const double (&v)[RPL]) {
#pragma unroll
    for (int r = 0; r < RPL; ++r) lds[slot * ldn + lane + 64 * r] = v[r];
  };
  auto lds_get = [&](int slot, double (&v)[RPL]) {
#pragma unroll
    for (int r = 0; r < RPL; ++r) v[r] = lds[slot * ldn + lane + 64 * r];
  };

  int sweeps_done = -1;
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    float mymax = 0.f;
    // ---- pairs inside each home block (blocks 2g and 2g+1), round robin over B columns ----
    for (int hb = 0; hb < 2; ++hb) {
      const int blk = 2 * g + hb;
      double mine[RPL];
      load_col(blk * B + w, mine);
      lds_put(w, mine);
      __syncthreads();
      for (int r = 0; r < B - 1; ++r) {
        if (w < B / 2) {
          int u, v;
          if (w == 0) {
            u = r % (B - 1);
            v = B - 1;
          } else {
            u = (r + w) % (B - 1);
            v = (r - w + (B - 1)) % (B - 1);
          }
          if (blk * B + u < n && blk * B + v < n) {
            double ap[RPL], aq[RPL];
            lds_get(u, ap);
            lds_get(v, aq);
            float rel = rotate_pair<RPL>(ap, aq, (float)tol, null2);
            mymax = fmaxf(mymax, rel);
            lds_put(u, ap);
            lds_put(v, aq);
          }
        }
        __syncthreads();
      }
      lds_get(w, mine);
      store_col(blk * B + w, mine);
      __syncthreads();
    }
    ++phase;
    grid_barrier(bar, phase * nwg, nwg, fail);
    // ---- block-pair rounds ----
    for (int r = 0; r < nblk - 1; ++r) {
      int I, J;
      if (g == 0) {
        I = r % (nblk - 1);
        J = nblk - 1;
      } else {
        I = (r + g) % (nblk - 1);
        J = (r - g + (nblk - 1)) % (nblk - 1);
      }
      double ap[RPL];
      load_col(I * B + w, ap);
      {
        double tmp[RPL];
        load_col(J * B + w, tmp);
        lds_put(w, tmp);
      }
      __syncthreads();
      const bool pvalid = I * B + w < n;
      for (int s = 0; s < B; ++s) {
        const int v = (w + s) % B;
        if (pvalid && J * B + v < n) {
          double aq[RPL];
          lds_get(v, aq);
          float rel = rotate_pair<RPL>(ap, aq, (float)tol, null2);
          mymax = fmaxf(mymax, rel);
          lds_put(v, aq);
        }
        __syncthreads();
      }
      store_col(I * B + w, ap);
      {
        double tmp[RPL];
        lds_get(w, tmp);
        store_col(J * B + w, tmp);
      }
      if (r == nblk - 2) {   // publish this workgroup's sweep maximum before the last barrier
        if (lane == 0) wg_max[w] = mymax;
        __syncthreads();
        if (threadIdx.x == 0) {
          float m = 0.f;
          for (int i = 0; i < B; ++i) m = fmaxf(m, wg_max[i]);
          atomicMax(&cv[sweep], f2ord(m));
        }
      }
      ++phase;
      grid_barrier(bar, phase * nwg, nwg, fail);
    }
    const float smax = __uint_as_float(__hip_atomic_load(&cv[sweep], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (smax <= (float)tol) {
      sweeps_done = sweep + 1;
      break;
    }
  }
  // ---- eigenvalues = column norms; rank them (descending) and emit unit eigenvectors ----
  double* norms = norms_all + (size_t)prob * nblk * B;
  for (int hb = 0; hb < 2; ++hb) {
    const int col = (2 * g + hb) * B + w;
    double mine[RPL];
    load_col(col, mine);
    double s = 0;
#pragma unroll
    for (int r = 0; r < RPL; ++r) s += mine[r] * mine[r];
    s = sqrt(wave_sum(s));
    if (lane == 0) st_shared(&norms[col], (col < n) ? s : -1.0);
  }
  ++phase;
  grid_barrier(bar, phase * nwg, nwg, fail);
  double* evals = evals_all + (size_t)prob * n;
  double* evecs = evecs_all + (size_t)prob * n * n;
  for (int hb = 0; hb < 2; ++hb) {
    const int col = (2 * g + hb) * B + w;
    if (col >= n) continue;
    const double me = ld_shared(&norms[col]);
    int cnt = 0;
    for (int j = lane; j < n; j += 64) {
      const double o = ld_shared(&norms[j]);
      cnt += (o > me || (o == me && j < col)) ? 1 : 0;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m, 64);
    double mine[RPL];
    load_col(col, mine);
    // sign: component of largest magnitude made positive (deterministic output)
    double best = -1.0, bval = 0.0;
    int bidx = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      int i = lane + 64 * r;
      double a = fabs(mine[r]);
      if (i < n && a > best) {
        best = a;
        bval = mine[r];
        bidx = i;
      }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      double ob = __shfl_xor(best, m, 64), ov = __shfl_xor(bval, m, 64);
      int oi = __shfl_xor(bidx, m, 64);
      if (ob > best || (ob == best && oi < bidx)) {
        best = ob;
        bval = ov;
        bidx = oi;
      }
    }
    const double sc = (me > 0.0) ? ((bval < 0 ? -1.0 : 1.0) / me) : 0.0;
    if (lane == 0) evals[cnt] = me;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      int i = lane + 64 * r;
      if (i < n) evecs[(size_t)cnt * n + i] = mine[r] * sc;
    }
  }
  if (g == 0 && threadIdx.x == 0) {
    info[prob] = sweeps_done;
    if (sweeps_done < 0) atomicAdd(fail, 1);   // sticky: read by vipmi_check_deferred
  }
}

// null2[prob] = (nulltol * largest column norm)^2, nulltol = max(1e-13, 4e-16 n): one workgroup per problem
__global__ __launch_bounds__(256) void jacobi_null_kernel(const double* __restrict__ Gall, int n, double* __restrict__ null2) {
  __shared__ double sh[4];
  const double* A = Gall + (size_t)blockIdx.x * n * n;
  double best = 0.0;
  for (int col = threadIdx.x >> 6; col < n; col += 4) {          // one wave per column (G symmetric: column = row, contiguous)
    double s = 0.0;
    for (int i = threadIdx.x & 63; i < n; i += 64) {
      const double v = A[(size_t)col * n + i];
      s += v * v;
    }
    s = wave_sum(s);
    best = s > best ? s : best;
  }
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = sh[0] > sh[1] ? sh[0] : sh[1];
    m = m > sh[2] ? m : sh[2];
    m = m > sh[3] ? m : sh[3];
    double nt = 4e-16 * n;
    if (nt < 1e-13) nt = 1e-13;
    null2[blockIdx.x] = nt * nt * m;
  }
}

template <int B, int RPL>
int launch_jacobi(vipmi_ctx* ctx, double* G, int64_t batch, int n, double* evals, double* evecs) {
  int nblk = (int)cdiv(n, B);
  if (nblk < 2) nblk = 2;
  if (nblk & 1) ++nblk;
  const int nwg = nblk / 2;
  const int max_sweeps = (int)ctx->opt("eigh_max_sweeps", 40);
  const double tol = 1e-12;
  const size_t lds_bytes = (size_t)B * RPL * 64 * sizeof(double);
  auto kern = jacobi_kernel<B, RPL>;
  VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds_bytes));
  // co-residency: workgroups per CU limited by LDS and by 2048 threads/CU
  int per_cu_lds = (int)((160 * 1024) / (lds_bytes + 256));
  int per_cu_thr = 2048 / (64 * B);
  int per_cu = per_cu_lds < per_cu_thr ? per_cu_lds : per_cu_thr;
  if (per_cu < 1) per_cu = 1;
  int64_t resident = (int64_t)per_cu * ctx->num_cu;
  if (nwg > resident) {
    set_error("eigh: n=%d needs %d co-resident workgroups (> %ld)", n, nwg, (long)resident);
    return VIPMI_ERR_UNSUPPORTED;
  }
  int64_t chunk = resident / nwg;
  if (chunk > 65535) chunk = 65535;
  unsigned* bars = nullptr;
  unsigned* conv = nullptr;
  int* info = nullptr;
  double* norms = nullptr;
  VIPMI_TRY(ws(ctx, "eigh_bars", (size_t)batch, &bars));
  VIPMI_TRY(ws(ctx, "eigh_conv", (size_t)batch * max_sweeps, &conv));
  VIPMI_TRY(ws(ctx, "eigh_info", (size_t)batch, &info));
  VIPMI_TRY(ws(ctx, "eigh_norms", (size_t)batch * nblk * B, &norms));
  int* fail = nullptr;
  VIPMI_TRY(deferred_fail_words(ctx, &fail));
  double* null2 = nullptr;
  VIPMI_TRY(ws(ctx, "eigh_null2", (size_t)batch, &null2));
  hipLaunchKernelGGL(jacobi_null_kernel, dim3((unsigned)batch), dim3(256), 0, ctx->stream, G, n, null2);
  VIPMI_CHECK_HIP(hipGetLastError());
  VIPMI_CHECK_HIP(hipMemsetAsync(bars, 0, sizeof(unsigned) * batch, ctx->stream));
  VIPMI_CHECK_HIP(hipMemsetAsync(conv, 0, sizeof(unsigned) * batch * max_sweeps, ctx->stream));
  VIPMI_CHECK_HIP(hipMemsetAsync(info, 0xff, sizeof(int) * batch, ctx->stream));
  for (int64_t p0 = 0; p0 < batch; p0 += chunk) {
    int64_t nb = batch - p0 < chunk ? batch - p0 : chunk;
    hipLaunchKernelGGL(kern, dim3(nwg, (unsigned)nb), dim3(64 * B), lds_bytes, ctx->stream, G, n, nblk,
                       max_sweeps, tol, bars, conv, info, evals, evecs, norms, (int)p0, fail, null2);
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  if (ctx->opt("eigh_check", 1)) {
    std::vector<int> h(batch);
    VIPMI_CHECK_HIP(hipMemcpyAsync(h.data(), info, sizeof(int) * batch, hipMemcpyDeviceToHost, ctx->stream));
    VIPMI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    for (int64_t i = 0; i < batch; ++i)
      if (h[i] < 0) {
        set_error("eigh: problem %ld did not converge in %d sweeps", (long)i, max_sweeps);
        return VIPMI_ERR_NOCONV;
      }
    ctx->options["eigh_last_sweeps"] = h[0];
  }
  return VIPMI_OK;
}

}  // namespace

int eigh_f64(vipmi_ctx* ctx, double* G, int64_t batch, int64_t n, double* evals, double* evecs) {
  VIPMI_REQUIRE(G && evals && evecs, "eigh: null pointer");
  VIPMI_REQUIRE(batch > 0 && n > 0, "eigh: bad sizes batch=%ld n=%ld", (long)batch, (long)n);
  StageScope sc(ctx, "eigh");
  const int rpl = (int)cdiv(n, 64);
  if (rpl <= 1) return launch_jacobi<16, 1>(ctx, G, batch, (int)n, evals, evecs);
  if (rpl <= 2) return launch_jacobi<16, 2>(ctx, G, batch, (int)n, evals, evecs);
  if (rpl <= 4) return launch_jacobi<16, 4>(ctx, G, batch, (int)n, evals, evecs);
  if (rpl <= 7) return launch_jacobi<16, 7>(ctx, G, batch, (int)n, evals, evecs);
  if (rpl <= 10) return launch_jacobi<16, 10>(ctx, G, batch, (int)n, evals, evecs);
  // larger problems: 8 waves per workgroup (256-VGPR budget, 8 columns x n doubles of LDS)
  if (rpl <= 16) return launch_jacobi<8, 16>(ctx, G, batch, (int)n, evals, evecs);
  if (rpl <= 32) return launch_jacobi<8, 32>(ctx, G, batch, (int)n, evals, evecs);
  set_error("eigh: matrices of more than 2048 x 2048 (cubes / libraries of more than 2048 frames) are not supported by "
            "the device eigensolvers yet (n=%ld)", (long)n);
  return VIPMI_ERR_UNSUPPORTED;
}

}  // namespace vipmi
