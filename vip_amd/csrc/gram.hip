// gram.hip -- C[na,nb] (float64) = A[na,P] * B[nb,P]^T on the CDNA4 matrix cores.
//
// Replaces the covariance product of svd_wrapper(mode='eigen') (psfsub/svd.py:449,
// `C = np.dot(matrix, matrix.T)`) and, through the Gram identity, the thin SVD of
// mode='lapack' (svd.py:470).  A == B gives the symmetric Gram matrix (upper block triangle
// computed once, mirrored by the reducer).
//
// Design (MFMA-bound: AI = n/2 flop/B at n frames):
//  * the contraction index is the PIXEL axis (P ~ 2.6e5), the output is tiny (n x n), so the
//    kernel is split-K: grid.x = K-slices, every workgroup owns one slice of pixels and computes
//    (a group of) all output tiles for it; partial tiles go to a scratch buffer and a second
//    kernel sums the slices in float64 in a fixed order (deterministic, no atomics).
//  * one wave owns a TB x TB super-tile of 16x16 MFMA blocks and keeps it in accumulator
//    registers for the whole slice.  Operands are loaded straight from global memory in MFMA
//    fragment layout with 16-byte loads: lane (r = lane&15, kq = lane>>4) loads
//    A[row0 + r][k0 + 4*kq .. +3]; component c of that float4 feeds MFMA number c.  Because both
//    operands use the same k permutation the contraction is unchanged, and no LDS staging,
//    barrier or transpose is needed (each load instruction covers 16 rows x 64 contiguous bytes).
//    All waves of a workgroup walk the same pixel slice, so rows are shared through L1/L2.
//  * accumulation: v_mfma_f64_16x16x4_f64 (inputs converted f32->f64, exact products, f64 sums)
//    by default -- the Gram matrix squares the condition number, so it is kept at f64 accuracy;
//    option "gram_f32" selects v_mfma_f32_16x16x4_f32 (2x MFMA rate, f32 chains per slice).
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace vipmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <bool VEC>
__device__ __forceinline__ f32x4 load_frag(const float* __restrict__ row, bool row_ok, int64_t off,
                                           int64_t P) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (!row_ok) return v;
  if (VEC && off + 4 <= P) {
    v = *reinterpret_cast<const f32x4*>(row + off);
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (off + c < P) v[c] = row[off + c];
  }
  return v;
}

// tiles: int2 (ti, tj) list.  partial layout: [slice][tile][bi][bj][256] (Acc type),
// element index inside a block = row*16 + col.
template <int TB>
constexpr int gram_max_waves() { return TB <= 2 ? 16 : 8; }

template <int TB, bool ACC64, bool VEC>
__global__ __launch_bounds__(64 * gram_max_waves<TB>()) void gram_partial_kernel(
    const float* __restrict__ A, const float* __restrict__ B, int na, int nb, int64_t P, int64_t ld,
    const int2* __restrict__ tiles, int ntiles, int waves_per_wg, int64_t klen, int symmetric,
    void* __restrict__ partial_, int64_t batch_stride, int nslices) {
  // blockIdx.z = problem of a batch (same shapes; operands batch_stride elements apart)
  A += (int64_t)blockIdx.z * batch_stride;
  B += (int64_t)blockIdx.z * batch_stride;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // in an SGPR: the tile guards are scalar branches
  const int tile = blockIdx.y * waves_per_wg + wave;
  if (tile >= ntiles) return;
  const int slice = blockIdx.x;
  const int2 t = tiles[tile];
  if (t.x < 0) return;                      // padding entry of the balanced tile order
  const int r = lane & 15, kq = lane >> 4;
  const int64_t kbeg = (int64_t)slice * klen;
  int64_t kend = kbeg + klen;
  if (kend > P) kend = P;

  const float* pa[TB];
  const float* pb[TB];
  bool oka[TB], okb[TB];
  bool blka[TB], blkb[TB];
#pragma unroll
  for (int i = 0; i < TB; ++i) {
    int ra = (t.x * TB + i) * 16 + r;
    int rb = (t.y * TB + i) * 16 + r;
    oka[i] = ra < na;
    okb[i] = rb < nb;
    blka[i] = (t.x * TB + i) * 16 < na;   // wave-uniform
    blkb[i] = (t.y * TB + i) * 16 < nb;
    pa[i] = A + (int64_t)(oka[i] ? ra : 0) * ld;
    pb[i] = B + (int64_t)(okb[i] ? rb : 0) * ld;
  }
  const bool diag = symmetric && (t.x == t.y);

  using acc_t = typename std::conditional<ACC64, f64x4, f32x4>::type;
  acc_t acc[TB][TB];
#pragma unroll
  for (int i = 0; i < TB; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

  f32x4 fa[TB], fb[TB];
  // Fast path: every block of the super-tile is live, it is not on the diagonal, and the slice is whole 16-pixel steps of
  // aligned 16-byte loads -- no guards at all: TB*TB MFMAs per step in one basic block (with the guards, which the
  // compiler turns into an exec-mask save / branch / restore around every MFMA, the matrix pipe was busy 55 % of the time)
  bool full = VEC && kbeg < kend && ((kend - kbeg) & 15) == 0;
#pragma unroll
  for (int i = 0; i < TB; ++i) full = full && (t.x * TB + i) * 16 + 15 < na && (t.y * TB + i) * 16 + 15 < nb;
  auto fast = [&](auto diag_c) {
    constexpr bool DG = decltype(diag_c)::value;           // diagonal super-tile: blocks j < i are skipped (compile time)
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      fa[i] = *reinterpret_cast<const f32x4*>(pa[i] + kbeg + 4 * kq);
      fb[i] = *reinterpret_cast<const f32x4*>(pb[i] + kbeg + 4 * kq);
    }
    for (int64_t k0 = kbeg; k0 < kend; k0 += 16) {
      f32x4 na_[TB], nb_[TB];
      const int64_t kn = (k0 + 16 < kend) ? k0 + 16 : k0;        // (the last step re-reads its own fragment: no branch)
#pragma unroll
      for (int i = 0; i < TB; ++i) {
        na_[i] = *reinterpret_cast<const f32x4*>(pa[i] + kn + 4 * kq);
        nb_[i] = *reinterpret_cast<const f32x4*>(pb[i] + kn + 4 * kq);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int i = 0; i < TB; ++i) {
#pragma unroll
          for (int j = 0; j < TB; ++j) {
            if (DG && j < i) continue;
            if constexpr (ACC64) {
              acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[i][c], (double)fb[j][c], acc[i][j], 0, 0, 0);
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][c], fb[j][c], acc[i][j], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < TB; ++i) {
        fa[i] = na_[i];
        fb[i] = nb_[i];
      }
    }
  };
  if (full && !diag) {
    fast(std::false_type{});
  } else if (full) {
    fast(std::true_type{});
  } else {
  if (kbeg < kend) {
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      fa[i] = load_frag<VEC>(pa[i], oka[i], kbeg + 4 * kq, kend);
      fb[i] = load_frag<VEC>(pb[i], okb[i], kbeg + 4 * kq, kend);
    }
  }
  for (int64_t k0 = kbeg; k0 < kend; k0 += 16) {
    f32x4 na_[TB], nb_[TB];
    const int64_t kn = k0 + 16;
    if (kn < kend) {
#pragma unroll
      for (int i = 0; i < TB; ++i) {
        na_[i] = load_frag<VEC>(pa[i], oka[i], kn + 4 * kq, kend);
        nb_[i] = load_frag<VEC>(pb[i], okb[i], kn + 4 * kq, kend);
      }
    } else {
#pragma unroll
      for (int i = 0; i < TB; ++i) {
        na_[i] = f32x4{0, 0, 0, 0};
        nb_[i] = f32x4{0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int i = 0; i < TB; ++i) {
        if (!blka[i]) continue;
#pragma unroll
        for (int j = 0; j < TB; ++j) {
          if (!blkb[j]) continue;
          if (diag && j < i) continue;
          if constexpr (ACC64) {
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fa[i][c], (double)fb[j][c],
                                                             acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][c], fb[j][c], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      fa[i] = na_[i];
      fb[i] = nb_[i];
    }
  }
  }   // guarded path

  using sc_t = typename std::conditional<ACC64, double, float>::type;
  sc_t* partial = reinterpret_cast<sc_t*>(partial_) +
                  (((int64_t)blockIdx.z * nslices + slice) * ntiles + tile) * (int64_t)(TB * TB * 256);
  const int col = lane & 15;
#pragma unroll
  for (int i = 0; i < TB; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      if (!blka[i] || !blkb[j] || (diag && j < i)) continue;
      sc_t* blk = partial + (i * TB + j) * 256;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // f64 16x16x4: row = (lane>>4) + 4*g ; f32 16x16x4: row = (lane>>4)*4 + g
        int row = ACC64 ? ((lane >> 4) + 4 * g) : ((lane >> 4) * 4 + g);
        blk[row * 16 + col] = acc[i][j][g];
      }
    }
}

// COOP adjacent lanes share an output element: lane q sums the slices q, q + COOP, ... and the lanes are combined by a fixed
// butterfly (deterministic).  Small problems only: one thread per element is a chain of `nslices` dependent loads on a few
// thousand threads (62 us for the 50 x 16384 Gram of C1, 5 times the partial products themselves).
template <int TB, bool ACC64, int COOP>
__global__ void gram_reduce_coop_kernel(const void* __restrict__ partial_, const int2* __restrict__ tiles,
                                        int ntiles, int nslices, int na, int nb, int symmetric,
                                        double* __restrict__ G) {
  using sc_t = typename std::conditional<ACC64, double, float>::type;
  const int64_t per_tile = TB * TB * 256;
  const int64_t total = (int64_t)ntiles * per_tile;
  const sc_t* partial = reinterpret_cast<const sc_t*>(partial_) + (int64_t)blockIdx.y * nslices * total;   // batch
  G += (int64_t)blockIdx.y * na * nb;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // (grid covers total * COOP threads exactly or more)
  const int64_t e = gid / COOP;
  const int q = (int)(gid % COOP);
  double s = 0.0;
  if (e < total)
    for (int sl = q; sl < nslices; sl += COOP) s += (double)partial[(int64_t)sl * total + e];
#pragma unroll
  for (int off = 1; off < COOP; off <<= 1) s += __shfl_xor(s, off, 64);
  if (e >= total || q != 0) return;
  const int tile = (int)(e / per_tile);
  const int rem = (int)(e % per_tile);
  const int blk = rem >> 8, idx = rem & 255;
  const int bi = blk / TB, bj = blk % TB;
  const int2 t = tiles[tile];
  if (t.x < 0) return;
  const int gi = (t.x * TB + bi) * 16 + (idx >> 4);
  const int gj = (t.y * TB + bj) * 16 + (idx & 15);
  if (gi >= na || gj >= nb) return;
  if (symmetric && t.x == t.y && bj < bi) return;
  G[(int64_t)gi * nb + gj] = s;
  if (symmetric) G[(int64_t)gj * nb + gi] = s;
}

template <int TB, bool ACC64>
__global__ void gram_reduce_kernel(const void* __restrict__ partial_, const int2* __restrict__ tiles,
                                   int ntiles, int nslices, int na, int nb, int symmetric,
                                   double* __restrict__ G) {
  using sc_t = typename std::conditional<ACC64, double, float>::type;
  const int64_t per_tile = TB * TB * 256;
  const int64_t total = (int64_t)ntiles * per_tile;
  const sc_t* partial = reinterpret_cast<const sc_t*>(partial_) + (int64_t)blockIdx.y * nslices * total;   // batch
  G += (int64_t)blockIdx.y * na * nb;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int tile = (int)(e / per_tile);
    int rem = (int)(e % per_tile);
    int blk = rem >> 8, idx = rem & 255;
    int bi = blk / TB, bj = blk % TB;
    int2 t = tiles[tile];
    if (t.x < 0) continue;
    int gi = (t.x * TB + bi) * 16 + (idx >> 4);
    int gj = (t.y * TB + bj) * 16 + (idx & 15);
    if (gi >= na || gj >= nb) continue;
    if (symmetric && t.x == t.y && bj < bi) continue;
    // (the slices in their order, sixteen loads in flight: a load per dependent add left this kernel at 76 us for a 400 x 400 matrix)
    double s = 0.0;
    int sl = 0;
    for (; sl + 16 <= nslices; sl += 16) {
      sc_t v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = partial[(int64_t)(sl + u) * total + e];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += (double)v[u];
    }
    for (; sl < nslices; ++sl) s += (double)partial[(int64_t)sl * total + e];
    G[(int64_t)gi * nb + gj] = s;
    if (symmetric) G[(int64_t)gj * nb + gi] = s;
  }
}

template <int TB, bool ACC64>
static int launch(vipmi_ctx* ctx, const float* A, int64_t na, const float* B, int64_t nb, int64_t P,
                  int64_t ld, double* G, bool symmetric, int64_t batch = 1) {
  const int nba = (int)cdiv(na, 16), nbb = (int)cdiv(nb, 16);
  const int nta = (int)cdiv(nba, TB), ntb = (int)cdiv(nbb, TB);
  std::vector<int2> tiles;
  for (int i = 0; i < nta; ++i)
    for (int j = symmetric ? i : 0; j < ntb; ++j) tiles.push_back(int2{i, j});
  // waves per workgroup: the accumulator tile caps TB>=3 kernels at 2 waves/SIMD
  int max_waves = gram_max_waves<TB>();
  {
    // four waves per workgroup (one per SIMD) instead of eight: finer-grained dispatch of the unequal tile groups
    // (C2 1.03 -> 0.98 ms, C5 78.8 -> 76.6 ms)
    const int o = (int)ctx->opt("gram_wpw", (TB >= 3 && batch == 1 && tiles.size() >= 16) ? 4 : 0);
    if (o >= 1 && o < max_waves) max_waves = o;
  }
  int wpw = (int)tiles.size() < max_waves ? (int)tiles.size() : max_waves;
  int ngroups = (int)cdiv((int64_t)tiles.size(), wpw);
  // Balance the MFMA work over the SIMDs: tiles differ (diagonal tiles skip their lower blocks, edge tiles their
  // padding), a workgroup holds its CU until its busiest SIMD is done, and wave w runs on SIMD w % 4.  Longest-
  // processing-time assignment of the tiles to (workgroup, SIMD) slots, slots sorted by load so that the four
  // slots of a workgroup are alike; unfilled positions become (-1, -1) entries that exit at once.
  if (wpw % 4 == 0 && ngroups >= 1 && (int)tiles.size() > 4) {
    auto work = [&](const int2& t) {
      int w = 0;
      for (int i = 0; i < TB; ++i)
        for (int j = 0; j < TB; ++j) {
          if ((t.x * TB + i) >= nba || (t.y * TB + j) >= nbb) continue;
          if (symmetric && t.x == t.y && j < i) continue;
          ++w;
        }
      return w;
    };
    const int q = wpw / 4, nslots = ngroups * 4;
    std::vector<int> order(tiles.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return work(tiles[a]) > work(tiles[b]); });
    std::vector<std::vector<int>> slot(nslots);
    std::vector<int> load(nslots, 0);
    for (int idx : order) {
      int best = -1;
      for (int sl = 0; sl < nslots; ++sl)
        if ((int)slot[sl].size() < q && (best < 0 || load[sl] < load[best])) best = sl;
      slot[best].push_back(idx);
      load[best] += work(tiles[idx]);
    }
    std::vector<int> sorder(nslots);
    for (int i = 0; i < nslots; ++i) sorder[i] = i;
    std::stable_sort(sorder.begin(), sorder.end(), [&](int a, int b) { return load[a] > load[b]; });
    std::vector<int2> balanced((size_t)ngroups * wpw, int2{-1, -1});
    for (int si = 0; si < nslots; ++si) {
      const int g = si / 4, simd = si % 4;
      const std::vector<int>& members = slot[sorder[si]];
      for (size_t pos = 0; pos < members.size(); ++pos) balanced[(size_t)g * wpw + simd + 4 * pos] = tiles[members[pos]];
    }
    tiles.swap(balanced);
  }
  const int ntiles = (int)tiles.size();
  // slices: ~2 workgroups per CU in total, slice length a multiple of 16, >= 64 pixels
  int64_t target = ctx->opt("gram_slices", 0);
  if (target <= 0) {
    // two workgroups fit a CU: slices so that the launch fills whole rounds of 2 * num_cu workgroups.  With many tile
    // groups (n = 2000: 66 groups) the plain ceiling gave 8 slices = 528 workgroups = one full round + a round of 16
    // workgroups, i.e. half of the MFMA time idle; search the neighbourhood for the best-filled last round.
    const int64_t slots = (int64_t)(wpw == 4 ? 4 : 2) * ctx->num_cu;       // workgroups that fit the chip at once (16 waves per CU)
    target = cdiv(slots, ngroups);
    if (batch == 1 && ngroups > 8) {
      double best = 0.0;
      int64_t pick = target;
      for (int64_t ns = target > 2 ? target / 2 : 1; ns <= 4 * target; ++ns) {
        const int64_t wgs = ns * ngroups;
        const double eff = (double)wgs / (double)(cdiv(wgs, slots) * slots);
        if (eff > best + 0.02) {           // prefer fewer slices (less partial traffic) unless clearly better filled
          best = eff;
          pick = ns;
        }
      }
      target = pick;
    }
  }
  if (batch > 1 && ctx->opt("gram_slices", 0) <= 0) {
    // many small problems: ~6 workgroups per CU in total, so that the hardware dispatcher evens out workgroups of
    // different weight (the tile groups of one problem are not alike)
    target = cdiv((int64_t)6 * ctx->num_cu, batch * ngroups);
    if (target < 1) target = 1;
  }
  int64_t klen = cdiv(cdiv(P, target), 16) * 16;
  if (klen < 64) klen = 64;
  int nslices = (int)cdiv(P, klen);
  if (nslices > 8 && !(batch == 1 && ngroups > 8 && ctx->opt("gram_slices", 0) <= 0)) {   // keep same-slice workgroups on one XCD (b % 8)
    nslices = (nslices / 8) * 8;
    klen = cdiv(cdiv(P, nslices), 16) * 16;
    nslices = (int)cdiv(P, klen);
  }
  int2* d_tiles = nullptr;
  {
    char key[96];
    snprintf(key, sizeof key, "%d/%d/%d/%d", TB, nta, ntb, (int)symmetric);
    void* p = nullptr;
    VIPMI_TRY(ctx->upload_cached("gram_tiles", key, tiles.data(), sizeof(int2) * ntiles, &p));
    d_tiles = reinterpret_cast<int2*>(p);
  }
  const size_t esz = ACC64 ? 8 : 4;
  void* partial = nullptr;
  VIPMI_TRY(ctx->get("gram_partial", (size_t)batch * nslices * ntiles * TB * TB * 256 * esz, &partial));
  // blocks that are skipped (out of range / lower triangle) are never read by the reducer
  const bool vec = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
  VIPMI_REQUIRE(batch <= 65535, "gram: batch too large");
  dim3 grid(nslices, ngroups, (unsigned)batch), block(64 * wpw);
  const int64_t bstride = na * ld;          // (batched: A == B, problems stacked row-wise)
  if (vec)
    hipLaunchKernelGGL((gram_partial_kernel<TB, ACC64, true>), grid, block, 0, ctx->stream, A, B,
                       (int)na, (int)nb, P, ld, d_tiles, ntiles, wpw, klen, (int)symmetric, partial, bstride, nslices);
  else
    hipLaunchKernelGGL((gram_partial_kernel<TB, ACC64, false>), grid, block, 0, ctx->stream, A, B,
                       (int)na, (int)nb, P, ld, d_tiles, ntiles, wpw, klen, (int)symmetric, partial, bstride, nslices);
  VIPMI_CHECK_HIP(hipGetLastError());
  int64_t total = (int64_t)ntiles * TB * TB * 256;
  if (total * batch <= 65536 && nslices >= 32) {          // small outputs, long slice chains: 8 lanes per element
    hipLaunchKernelGGL((gram_reduce_coop_kernel<TB, ACC64, 8>), dim3((unsigned)cdiv(total * 8, 256), (unsigned)batch), dim3(256),
                       0, ctx->stream, partial, d_tiles, ntiles, nslices, (int)na, (int)nb, (int)symmetric, G);
    VIPMI_CHECK_HIP(hipGetLastError());
    return VIPMI_OK;
  }
  int rb = (int)cdiv(total, 256);
  if (rb > 4096) rb = 4096;
  if (batch > 1 && rb > 64) rb = 64;
  hipLaunchKernelGGL((gram_reduce_kernel<TB, ACC64>), dim3(rb, (unsigned)batch), dim3(256), 0, ctx->stream, partial,
                     d_tiles, ntiles, nslices, (int)na, (int)nb, (int)symmetric, G);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

int gram_f32(vipmi_ctx* ctx, const float* A, int64_t na, const float* B, int64_t nb, int64_t P,
             int64_t ld, double* G) {
  VIPMI_REQUIRE(A && B && G, "gram: null pointer");
  VIPMI_REQUIRE(na > 0 && nb > 0 && P > 0 && ld >= P, "gram: bad sizes na=%ld nb=%ld P=%ld ld=%ld",
                (long)na, (long)nb, (long)P, (long)ld);
  VIPMI_REQUIRE(na < (1 << 20) && nb < (1 << 20), "gram: too many rows");
  StageScope sc(ctx, "gram");
  const bool symmetric = (A == B) && (na == nb);
  const bool f32acc = ctx->opt("gram_f32", 0) != 0;
  {
    // int8 matrix cores (gram_i8.hip): option gram_i8 = 0 off, 1 / 2 forced (5 / 6 digits), unset = where it is measured to
    // pay (tools/gram_i8_sizes.py: 1.3-1.5x from 256 rows and 2^25 elements, 1.9x at 2000 x 1024^2; slower than this kernel on
    // small or skinny problems)
    const int64_t i8 = ctx->opt("gram_i8", -1);
    const bool forced = i8 > 0 && na >= ctx->opt("gram_i8_min_n", 32) && P >= 1024;
    const bool pays = i8 < 0 && na >= 256 && P >= 32768 && na * P >= ((int64_t)1 << 25);
    if (symmetric && !f32acc && (forced || pays)) {
      // the digit planes need batch * S * npad * Ppad bytes (10.7 GB at 2000 x 1024^2) PER CONTEXT: when that allocation fails the
      // float64-MFMA kernel below, which needs none of it, serves the call as it did before the int8 path existed
      const int st = gram_i8_f32(ctx, A, na, P, ld, G, 1, i8 == 2 ? 2 : 1);
      if (st != VIPMI_ERR_NOMEM) return st;
      (void)hipGetLastError();                       // (the failed hipMalloc must not surface at the next launch check)
    }
  }
  const int64_t nmax = na > nb ? na : nb;
  int tb = (int)ctx->opt("gram_tb", 0);
  if (tb <= 0) tb = nmax <= 16 ? 1 : (nmax <= 32 ? 2 : 4);
  if (f32acc) {
    switch (tb) {
      case 1: return launch<1, false>(ctx, A, na, B, nb, P, ld, G, symmetric);
      case 2: return launch<2, false>(ctx, A, na, B, nb, P, ld, G, symmetric);
      case 3: return launch<3, false>(ctx, A, na, B, nb, P, ld, G, symmetric);
      case 5: return launch<5, false>(ctx, A, na, B, nb, P, ld, G, symmetric);
      default: return launch<4, false>(ctx, A, na, B, nb, P, ld, G, symmetric);
    }
  }
  switch (tb) {
    case 1: return launch<1, true>(ctx, A, na, B, nb, P, ld, G, symmetric);
    case 2: return launch<2, true>(ctx, A, na, B, nb, P, ld, G, symmetric);
    case 3: return launch<3, true>(ctx, A, na, B, nb, P, ld, G, symmetric);
    default: return launch<4, true>(ctx, A, na, B, nb, P, ld, G, symmetric);
  }
}

// `batch` symmetric Gram matrices of equal shape: M is [batch][n][P] (row length ld = P), G is [batch][n][n].
// One launch for all of them (ADI+mSDI: one 39 x 39 spectral Gram matrix per multispectral frame).
int gram_batched_f32(vipmi_ctx* ctx, const float* M, int64_t batch, int64_t n, int64_t P, double* G) {
  VIPMI_REQUIRE(M && G, "gram_batched: null pointer");
  VIPMI_REQUIRE(batch > 0 && n > 0 && P > 0 && n < (1 << 20), "gram_batched: bad sizes batch=%ld n=%ld P=%ld", (long)batch,
                (long)n, (long)P);
  StageScope sc(ctx, "gram");
  {
    // batches: the int8 path from 128 rows and 2^25 elements in all (tools/gram_i8_sizes.py: 39 x (200 x 65536) 3.0 -> 2.6 ms,
    // 8 x (196 x 30000) 0.43 -> 0.29 ms; 200 x (39 x 65536) and 16 x (100 x 16384) are faster on the float64 MFMA)
    const int64_t i8 = ctx->opt("gram_i8", -1);
    const bool fits = ctx->opt("gram_f32", 0) == 0 && P >= 1024 && batch <= 65535;
    const bool forced = i8 > 0 && n >= ctx->opt("gram_i8_min_n", 32);
    const bool pays = i8 < 0 && n >= 128 && P >= 16384 && batch * n * P >= ((int64_t)1 << 25);
    if (fits && (forced || pays)) {
      const int st = gram_i8_f32(ctx, M, n, P, P, G, batch, i8 == 2 ? 2 : 1);
      if (st != VIPMI_ERR_NOMEM) return st;          // (no room for the digit planes: the float64-MFMA kernel below)
      (void)hipGetLastError();
    }
  }
  for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
    const int64_t nb_ = (batch - b0) < 65535 ? (batch - b0) : 65535;
    const float* Mb = M + (size_t)b0 * n * P;
    double* Gb = G + (size_t)b0 * n * n;
    int tb = (int)ctx->opt("gram_tb", 0);
    if (tb <= 0) tb = n <= 16 ? 1 : (n <= 32 ? 2 : 4);
    if (tb == 1) VIPMI_TRY((launch<1, true>(ctx, Mb, n, Mb, n, P, P, Gb, true, nb_)));
    else if (tb == 2) VIPMI_TRY((launch<2, true>(ctx, Mb, n, Mb, n, P, P, Gb, true, nb_)));
    else if (tb == 3) VIPMI_TRY((launch<3, true>(ctx, Mb, n, Mb, n, P, P, Gb, true, nb_)));
    else VIPMI_TRY((launch<4, true>(ctx, Mb, n, Mb, n, P, P, Gb, true, nb_)));
  }
  return VIPMI_OK;
}

}  // namespace vipmi
