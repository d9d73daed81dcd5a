// gram_i8.hip -- the Gram matrix G = M M^T of a float32 matrix on the INT8 matrix cores (v_mfma_i32_16x16x64_i8, ~4 POPS on
// MI355X: 50x the float64 MFMA rate), to float64 accuracy: the "Ozaki scheme" with integer slices.
//
// psfsub/svd.py:449 forms the covariance in float64 because it squares the condition number; gram.hip does so on
// v_mfma_f64_16x16x4_f64 (78.6 TF/s peak, 66 % reached: 1.08 ms at C2, 77 ms at C5 -- 19 % / 37 % of those calls).  Here:
//   1. split (gram_split_kernel): every row is cut into K-slices (the split-K slices of the product); per (row, slice) the
//      samples become fixed-point numbers with T = 7 S bits below 2^e, e = exponent of the slice's largest |a|,
//      written as S signed (balanced, |d| <= 64) base-128 digits d_0 (least significant) .. d_{S-1} in S int8 planes.  A local exponent per slice
//      keeps faint regions of a frame as accurate as bright ones.
//   2. product (gram_i8_kernel): a wave owns a 32 x 32 tile of G and a slice; digit planes j (rows) and j' (columns) are
//      multiplied on the int8 MFMA with EXACT int32 accumulation over the slice (|d| <= 64, top digit <= 127: no overflow up to
//      16384 samples), one accumulator set per significance level L = j + j'; levels below S - 1 - KEEP are dropped (their
//      terms are < 2^-7(S+1-KEEP) of the largest).  At the end of the slice the levels are combined in float64,
//      sum_L 128^L acc_L, scaled by the two rows' 2^(e+2-T), and written as a float64 partial tile;
//   3. the slices are summed in float64 in a fixed order (deterministic).
// Accuracy against an exact float64 product (numpy prototype and tools/gram_parity.py): S = 5, KEEP = 1 (19 digit products):
// max |dG| = 7e-12 max|G|;  S = 6, KEEP = 1 (26): 2e-15.  The exact float32 products of gram.hip give 1e-16: see DESIGN 3.4 for
// what 7e-12 means for the leading subspace (a rotation of 3e-7 between pairs 20 and 21 of C2).
// Non-finite samples make their row's scale NaN: the row and column of G are NaN, as with the float64 kernel.
#include "common.h"
#include <algorithm>
#include <vector>

namespace vipmi {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

// ---- 1. split: M[n][ld] float32 -> D[slice][row block][step][S][64 rows][64 samples] int8, sc[npad][nslices] float64 ----------
// The digits are stored in the order the product reads them: what a 64-row block needs for one 64-sample step -- S planes of
// 64 x 64 bytes -- is ONE contiguous 4 S KB piece, so every wave-wide load of the product covers 1 KB of whole cache lines.
// (Row-major planes [S][row][sample], the first layout, made each such load touch 16 rows at 64 of their line's 128 bytes: the
// product then asked its L2 for twice the bytes it used and sat at 43 % of the MFMA rate waiting for them.)
// One workgroup per (row, slice): the slice is read ONCE into registers (16 consecutive samples per thread and pass), its largest
// magnitude found through LDS, and the digits are peeled off in float32 -- y = a 2^(6-ex) (|y| < 64); d = rint(y); y = 128 (y - d);
// ... : every subtraction is exact (the remainder is a multiple of the sample's ulp and at most half a unit), the digits come out
// balanced in [-64, 64], and the last one carries the rounding.  a = sum_j d_j 128^j 2^(ex - 6 - 7 (S-1)) to 2^-(7S-1) of 2^ex.
template <int S, int NP>                           // NP = passes of 256 x 16 samples (slice length <= 4096 NP)
__global__ __launch_bounds__(256) void gram_split_kernel(const float* __restrict__ M, int n, int64_t P, int64_t ld,
                                                         int klen, int nslices, int64_t Ppad, int64_t plane,
                                                         int8_t* __restrict__ D, double* __restrict__ sc,
                                                         int64_t bstride_in, int npad, int row0) {
  // (blockIdx.y + row0 = row: the incremental front below splits one block of 64 rows at a time, as the rows arrive from the host)
  __shared__ float red[256];
  const int slice = blockIdx.x, row = blockIdx.y + row0, tid = threadIdx.x;
  M += (int64_t)blockIdx.z * bstride_in;
  D += (int64_t)blockIdx.z * S * plane;
  sc += (int64_t)blockIdx.z * npad * nslices;
  const int64_t k0 = (int64_t)slice * klen;
  const int nstep = klen >> 6;
  int8_t* drow = D + ((int64_t)slice * (npad >> 6) + (row >> 6)) * nstep * (S * 4096) + (row & 63) * 64;   // + tiled(e, j)
#define VIPMI_DIGIT_AT(e, j) (drow + ((int64_t)((e) >> 6) * S + (j)) * 4096 + ((e) & 63))
  if (row >= n) {                                   // padding rows: zero digits
    for (int e = tid * 16; e < klen; e += 256 * 16)
#pragma unroll
      for (int j = 0; j < S; ++j) *reinterpret_cast<v4i*>(VIPMI_DIGIT_AT(e, j)) = v4i{0, 0, 0, 0};
    if (tid == 0) sc[(int64_t)row * nslices + slice] = 0.0;
    return;
  }
  const float* src = M + (int64_t)row * ld + k0;
  const int64_t valid = (P - k0) < klen ? (P - k0) : klen;       // samples of this slice inside the row
  const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(M) & 15) == 0);
  float av[NP][16];
  float amax = 0.f;
  bool bad = false;
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int e = (ps * 256 + tid) * 16;
    if (vec && e + 16 <= valid) {
#pragma unroll
      for (int u4 = 0; u4 < 4; ++u4) {
        const float4 f = *reinterpret_cast<const float4*>(src + e + 4 * u4);
        av[ps][4 * u4] = f.x; av[ps][4 * u4 + 1] = f.y; av[ps][4 * u4 + 2] = f.z; av[ps][4 * u4 + 3] = f.w;
      }
    } else if (e + 16 <= valid) {        // rows of an odd length start at every alignment: 16-byte loads at 4-byte alignment
      typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
      for (int u4 = 0; u4 < 4; ++u4) {
        const f4u f = *reinterpret_cast<const f4u*>(src + e + 4 * u4);
        av[ps][4 * u4] = f[0]; av[ps][4 * u4 + 1] = f[1]; av[ps][4 * u4 + 2] = f[2]; av[ps][4 * u4 + 3] = f[3];
      }
    } else {
#pragma unroll
      for (int u = 0; u < 16; ++u) av[ps][u] = (e + u < valid) ? src[e + u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float a = fabsf(av[ps][u]);
      if (!(a <= 3.0e38f)) bad = true;              // NaN / Inf poison the slice
      amax = fmaxf(amax, a);
    }
  }
  red[tid] = bad ? __int_as_float(0x7fc00000) : amax;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      const float x = red[tid], y = red[tid + s];
      red[tid] = (x != x || y != y) ? __int_as_float(0x7fc00000) : fmaxf(x, y);
    }
    __syncthreads();
  }
  amax = red[0];
  const bool poisoned = amax != amax;
  int ex = -100;
  if (!poisoned && amax > 0.f) (void)frexpf(amax, &ex);          // amax = m 2^ex, 0.5 <= m < 1  ->  |a| < 2^ex
  if (ex < -100) ex = -100;                                        // (denormal slices: keep the scale factors finite in float32)
  const float up = ldexpf(1.f, 6 - ex);
  if (tid == 0)
    sc[(int64_t)row * nslices + slice] = poisoned ? __longlong_as_double(0x7ff8000000000000ll) : ldexp(1.0, ex - 6 - 7 * (S - 1));
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int e = (ps * 256 + tid) * 16;
    if (e >= klen) break;
    unsigned w[S][4];
#pragma unroll
    for (int j = 0; j < S; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) w[j][c] = 0u;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      float y = poisoned ? 0.f : av[ps][u] * up;
#pragma unroll
      for (int j = S - 1; j >= 0; --j) {              // most significant digit first
        const float d = rintf(y);
        w[j][u >> 2] |= ((unsigned)(int)d & 255u) << (8 * (u & 3));
        y = (y - d) * 128.f;
      }
    }
#pragma unroll
    for (int j = 0; j < S; ++j)
      *reinterpret_cast<v4i*>(VIPMI_DIGIT_AT(e, j)) = v4i{(int)w[j][0], (int)w[j][1], (int)w[j][2], (int)w[j][3]};
  }
#undef VIPMI_DIGIT_AT
}

// ---- 2. product: a workgroup of four waves = one 64 x 64 tile of G over one slice, every wave a 32 x 32 quarter ---------------
// Operands go through LDS: per 64-sample step the workgroup loads the S digit planes of its 64 rows and 64 columns ONCE
// (one 16-byte load per thread, plane and side; the loads of step i+1 are in flight in registers while the MFMAs of step i
// issue, then stored into the other half of the double buffer), every wave reads its two row blocks and two column blocks from
// there in MFMA fragment layout (lane (r, kq): 16 bytes of row r at sample offset 16 kq -- a 16 x 64-byte block is 1 KB
// contiguous: conflict-free ds_read_b128).  Straight from global memory (first version) the same product was bound by L1 / L2
// round trips: 1.38 ms at C2 against 0.25 ms of int8 MFMA time.
// the step loop of one workgroup, specialised at compile time: DIAG = the tile lies on the diagonal (its columns are its rows:
// one side is loaded), ROLE = 0 full quarter (four blocks), 1 quarter on the diagonal (block (1, 0) skipped), 2 idle quarter
// (below the diagonal: the wave only helps loading).  With these as run-time flags every MFMA sat in a basic block of its own
// behind a branch and a just-in-time s_waitcnt (the lesson of gram.hip's guarded tile loop, again).
// 16 bytes per lane straight from global memory into LDS (global_load_lds_dwordx4: destination = M0 + 16 * lane, wave-uniform
// base; no staging registers, no ds_write pass).  hipcc does not count this load: the caller waits with glds_wait().
__device__ __forceinline__ void glds16(const int8_t* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int S, int KEEP, bool DIAG, int ROLE, bool DMA>
__device__ __forceinline__ void gram_i8_steps(int8_t* smem, const int8_t* ga, const int8_t* gb, int nsteps, int nbuf,
                                              int loff, int fa, int fb, v4i (&acc)[2 * S - 1 - (S - 1 - KEEP)][2][2], int wave) {
  constexpr int LMIN = S - 1 - KEEP;
  constexpr int PL = 64 * 64;                       // one plane of one side: 64 rows of 64 bytes (an 80-byte row stride removes the
                                                    // bank conflicts of the fragment reads -- a third of the LDS cycles -- but changes
                                                    // nothing: 0.78 against 0.75 ms, and two buffers no longer fit twice per CU)
  constexpr int SIDE = S * PL, BUF = 2 * SIDE;
  constexpr int plane = PL;                         // (global layout = LDS layout: [step][plane][row][64 bytes])
  v4i la[S], lb[S];
  // DMA (two buffers only): wave w moves rows 16 w .. 16 w + 15 of every plane, 1 KB per instruction, into the same place the
  // register-staged path stores them
  const unsigned lds0 = (unsigned)(uintptr_t)smem + (unsigned)wave * 1024u;
  auto dma = [&](unsigned buf, int64_t o) {
#pragma unroll
    for (int j = 0; j < S; ++j) {
      glds16(ga + j * plane + o, lds0 + buf + j * PL);
      if (!DIAG) glds16(gb + j * plane + o, lds0 + buf + SIDE + j * PL);
    }
  };
  if (DMA) {
    dma(0u, 0);
    glds_wait();
  } else {
#pragma unroll
    for (int j = 0; j < S; ++j) {
      la[j] = *reinterpret_cast<const v4i*>(ga + j * plane);
      if (!DIAG) lb[j] = *reinterpret_cast<const v4i*>(gb + j * plane);
    }
#pragma unroll
    for (int j = 0; j < S; ++j) {
      *reinterpret_cast<v4i*>(smem + j * PL + loff) = la[j];
      if (!DIAG) *reinterpret_cast<v4i*>(smem + SIDE + j * PL + loff) = lb[j];
    }
  }
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int8_t* cur = smem + (nbuf == 2 ? (step & 1) * BUF : 0);
    int8_t* nxt = smem + (nbuf == 2 ? ((step + 1) & 1) * BUF : 0);
    const bool more = step + 1 < nsteps;
    if (more) {
      const int64_t o = (int64_t)(step + 1) * SIDE;
      if (DMA) {
        dma(((step + 1) & 1) * BUF, o);
      } else {
#pragma unroll
        for (int j = 0; j < S; ++j) {
          la[j] = *reinterpret_cast<const v4i*>(ga + j * plane + o);
          if (!DIAG) lb[j] = *reinterpret_cast<const v4i*>(gb + j * plane + o);
        }
      }
    }
    if (ROLE != 2) {
      // the column digits stay in registers for the step, the row digits are read plane by plane as they are used (the full
      // set of both sides made 312 VGPRs: one workgroup per CU, every LDS / global wait exposed)
      v4i b[2][S];
      const int8_t* cb = DIAG ? cur - SIDE : cur;       // a diagonal tile's columns are its rows: the A side serves both
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < S; ++j) b[i][j] = *reinterpret_cast<const v4i*>(cb + fb + i * 1024 + j * PL);
#pragma unroll
      for (int j = 0; j < S; ++j) {
        v4i a[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const v4i*>(cur + fa + i * 1024 + j * PL);
#pragma unroll
        for (int jp = 0; jp < S; ++jp) {
          const int l = j + jp - LMIN;
          if (l < 0) continue;
#pragma unroll
          for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 2; ++bj) {
              if (ROLE == 1 && bj < bi) continue;
              acc[l][bi][bj] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[bi], b[bj][jp], acc[l][bi][bj], 0, 0, 0);
            }
        }
      }
    }
    if (nbuf == 1) __syncthreads();                   // single buffer: everybody has read this step before it is overwritten
    if (DMA) {
      glds_wait();
    } else if (more) {
#pragma unroll
      for (int j = 0; j < S; ++j) {
        *reinterpret_cast<v4i*>(nxt + j * PL + loff) = la[j];
        if (!DIAG) *reinterpret_cast<v4i*>(nxt + SIDE + j * PL + loff) = lb[j];
      }
    }
    __syncthreads();
  }
}

template <int S, int KEEP, bool DMA>
__global__ __launch_bounds__(256, 2) void gram_i8_kernel(const int8_t* __restrict__ D, const double* __restrict__ sc, int npad,
                                                      int klen, int nslices, int64_t Ppad, int64_t plane,
                                                      const int2* __restrict__ wgtiles, int nwg,
                                                      double* __restrict__ partial, int gram_i8_nbuf, int wg0) {
  constexpr int LMIN = S - 1 - KEEP, NL = 2 * S - 2 - LMIN + 1;
  extern __shared__ __attribute__((aligned(16))) int8_t smem[];          // [1 or 2 buffers][2 sides][S planes][64 rows][64 bytes]
  constexpr int SIDE = S * 64 * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blockIdx.x = (slice % 8) + 8 * tile, blockIdx.y = slice / 8: workgroup ids go round-robin over the 8 XCDs, so every tile of
  // a slice runs on ONE XCD at the same time and the slice's digit rows are fetched once into that L2 (each row block is an
  // operand of n / 64 tiles; with the slices fastest the same product moved 3.7 GB from HBM: 0.99 ms)
  const int slice = (int)blockIdx.y * 8 + ((int)blockIdx.x & 7), wg = ((int)blockIdx.x >> 3) + wg0;     // wg0: first tile of a partial launch
  if (slice >= nslices) return;                      // (uniform for the workgroup: before any barrier)
  D += (int64_t)blockIdx.z * S * plane;
  sc += (int64_t)blockIdx.z * npad * nslices;
  const int2 t = wgtiles[wg];
  const bool diag = __builtin_amdgcn_readfirstlane((int)(t.x == t.y)) != 0;
  const int wi = wave >> 1, wj = wave & 1;
  // global -> LDS: thread -> (row = tid / 4, 16-byte segment = tid % 4) of every plane and side
  const int lrow = tid >> 2, lseg = tid & 3;
  const int64_t tile = (int64_t)(klen >> 6) * (S * 4096);          // bytes of one (slice, row block)
  const int8_t* ga = D + ((int64_t)slice * (npad >> 6) + t.x) * tile + 16 * tid;
  const int8_t* gb = D + ((int64_t)slice * (npad >> 6) + t.y) * tile + 16 * tid;
  const int loff = lrow * 64 + 16 * lseg;
  const int r = lane & 15, kq = lane >> 4;
  const int fa = (wi * 32 + r) * 64 + 16 * kq, fb = SIDE + (wj * 32 + r) * 64 + 16 * kq;      // fragment offsets of block 0
  v4i acc[NL][2][2];
#pragma unroll
  for (int l = 0; l < NL; ++l)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[l][i][j] = v4i{0, 0, 0, 0};
  const int nsteps = klen >> 6, nbuf = gram_i8_nbuf;
  int role = 0;
  if (!diag) {
    gram_i8_steps<S, KEEP, false, 0, DMA>(smem, ga, gb, nsteps, nbuf, loff, fa, fb, acc, wave);
  } else {
    role = wi > wj ? 2 : (wi == wj ? 1 : 0);          // (wave-uniform)
    if (role == 0) gram_i8_steps<S, KEEP, true, 0, DMA>(smem, ga, gb, nsteps, nbuf, loff, fa, fb, acc, wave);
    else if (role == 1) gram_i8_steps<S, KEEP, true, 1, DMA>(smem, ga, gb, nsteps, nbuf, loff, fa, fb, acc, wave);
    else gram_i8_steps<S, KEEP, true, 2, DMA>(smem, ga, gb, nsteps, nbuf, loff, fa, fb, acc, wave);
  }
  if (role == 2) return;
  const bool qdiag = role == 1;
  // combine the levels in float64, scale, write the quarter's partial tile: block (bi, bj) at (bi * 2 + bj) * 256, element
  // row * 16 + col (int8 / float32 MFMA output layout: row = 4 (lane >> 4) + g, column = lane & 15)
  double* out = partial + ((((int64_t)blockIdx.z * nslices + slice) * nwg + wg) * 4 + wave) * 1024;
  const double lm = (double)(1ll << (7 * LMIN));
  const int row0 = t.x * 64 + wi * 32, col0 = t.y * 64 + wj * 32;
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
      if (qdiag && bj < bi) continue;
      const double sb = sc[(int64_t)(col0 + bj * 16 + r) * nslices + slice] * lm;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = 4 * kq + g;
        const double sa = sc[(int64_t)(row0 + bi * 16 + row) * nslices + slice];
        double v = 0.0;
#pragma unroll
        for (int l = NL - 1; l >= 0; --l) v = v * 128.0 + (double)acc[l][bi][bj][g];     // Horner: sum_l 128^l acc_l
        out[(bi * 2 + bj) * 256 + row * 16 + r] = v * sa * sb;
      }
    }
}

// ---- 3. slices summed in a fixed order, upper triangle mirrored -------------------------------------------------------------
__global__ void gram_i8_reduce_kernel(const double* __restrict__ partial, const int2* __restrict__ wgtiles, int nwg,
                                      int nslices, int n, double* __restrict__ G) {
  const int64_t total = (int64_t)nwg * 4096;
  partial += (int64_t)blockIdx.y * nslices * total;
  G += (int64_t)blockIdx.y * n * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int wg = (int)(e >> 12), wave = (int)(e >> 10) & 3, rem = (int)(e & 1023);
    const int blk = rem >> 8, idx = rem & 255;
    const int bi = blk >> 1, bj = blk & 1, wi = wave >> 1, wj = wave & 1;
    const int2 t = wgtiles[wg];
    if (t.x == t.y && (wi > wj || (wi == wj && bj < bi))) continue;          // never written: lower part of a diagonal tile
    const int gi = t.x * 64 + wi * 32 + bi * 16 + (idx >> 4), gj = t.y * 64 + wj * 32 + bj * 16 + (idx & 15);
    if (gi >= n || gj >= n) continue;
    double s = 0.0;                         // (the slices in their order, sixteen loads in flight: see gram_reduce_kernel)
    int sl = 0;
    for (; sl + 16 <= nslices; sl += 16) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = partial[(int64_t)(sl + u) * total + e];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; sl + 4 <= nslices; sl += 4) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = partial[(int64_t)(sl + u) * total + e];
#pragma unroll
      for (int u = 0; u < 4; ++u) s += v[u];
    }
    for (; sl < nslices; ++sl) s += partial[(int64_t)sl * total + e];
    G[(int64_t)gi * n + gj] = s;
    G[(int64_t)gj * n + gi] = s;
  }
}

template <int S, int KEEP>
int run(vipmi_ctx* ctx, const float* M, int64_t n, int64_t P, int64_t ld, double* G, int64_t batch) {
  const int npad = (int)cdiv(n, 64) * 64, nt = npad / 64;
  std::vector<int2> tiles;
  // Tile order = dispatch order on every XCD (ids go round-robin over the XCDs, the slice is the id's low bits): ~64 consecutive
  // tiles of a slice are resident on an XCD at the same time and walk through the slice in step, so tiles listed together
  // should share operands -- 8 x 8 super-blocks (8 + 8 row blocks feed 64 tiles; row-major order re-fetched every column block
  // for every tile row: 351 GB through the L2s at C5).  Small problems (<= 8 tile rows): off-diagonal tiles first, the lighter
  // diagonal ones fill the tail.
  if (nt <= 8) {
    for (int i = 0; i < nt; ++i)
      for (int j = i + 1; j < nt; ++j) tiles.push_back(int2{i, j});
    for (int i = 0; i < nt; ++i) tiles.push_back(int2{i, i});
  } else {
    for (int I = 0; I < nt; I += 8)
      for (int J = I; J < nt; J += 8)
        for (int i = I; i < std::min(I + 8, nt); ++i)
          for (int j = std::max(J, i); j < std::min(J + 8, nt); ++j) tiles.push_back(int2{i, j});
  }
  const int nwg = (int)tiles.size();
  // slices: ~6 workgroups per CU in total (two are resident at a time: 80 KB of LDS each); a multiple of 64 samples, at most
  // 8192 (int32 head room: 5 x 8192 x 65 x 64 < 2^31)
  int64_t want = cdiv((int64_t)6 * ctx->num_cu, (int64_t)nwg * batch);
  if (ctx->opt("gram_i8_slices", 0) > 0) want = ctx->opt("gram_i8_slices", 0);
  if (want < 1) want = 1;
  int64_t klen = cdiv(cdiv(P, want), 64) * 64;
  if (klen < 256) klen = 256;
  if (klen > 8192) klen = 8192;
  const int nslices = (int)cdiv(P, klen);
  const int64_t Ppad = (int64_t)nslices * klen, plane = (int64_t)npad * Ppad;
  int8_t* D = nullptr;
  double *sc = nullptr, *partial = nullptr;
  VIPMI_TRY(ws(ctx, "gram_i8_digits", (size_t)batch * S * plane, &D));
  VIPMI_TRY(ws(ctx, "gram_i8_scale", (size_t)batch * npad * nslices, &sc));
  VIPMI_TRY(ws(ctx, "gram_i8_partial", (size_t)batch * nslices * nwg * 4096, &partial));
  int2* d_tiles = nullptr;
  {
    char key[64];
    snprintf(key, sizeof key, "i8/%d", nt);
    void* p = nullptr;
    VIPMI_TRY(ctx->upload_cached("gram_i8_tiles", key, tiles.data(), sizeof(int2) * nwg, &p));
    d_tiles = reinterpret_cast<int2*>(p);
  }
  VIPMI_REQUIRE(batch <= 65535, "gram: batch too large");
  {
    const dim3 sg((unsigned)nslices, (unsigned)npad, (unsigned)batch);
    if (klen <= 4096)
      hipLaunchKernelGGL((gram_split_kernel<S, 1>), sg, dim3(256), 0, ctx->stream, M, (int)n, P, ld, (int)klen, nslices, Ppad, plane,
                         D, sc, (int64_t)n * ld, npad, 0);
    else
      hipLaunchKernelGGL((gram_split_kernel<S, 2>), sg, dim3(256), 0, ctx->stream, M, (int)n, P, ld, (int)klen, nslices, Ppad, plane,
                         D, sc, (int64_t)n * ld, npad, 0);
  }
  VIPMI_CHECK_HIP(hipGetLastError());
  int nbuf = (int)ctx->opt("gram_i8_nbuf", 0);                            // LDS buffers per workgroup (0 = default: 5 digits 2 x 40 KB,
  if (nbuf != 1 && nbuf != 2) nbuf = S <= 5 ? 2 : 1;                        // two workgroups per CU; 6 digits one buffer)
  const size_t lds = (size_t)(nbuf == 2 ? 2 : 1) * 2 * S * 64 * 64;
  // global -> LDS by DMA (two buffers) unless gram_i8_dma = 0
  const bool dma = nbuf == 2 && ctx->opt("gram_i8_dma", 1) != 0;
  auto kern = dma ? gram_i8_kernel<S, KEEP, true> : gram_i8_kernel<S, KEEP, false>;
  VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * nwg), (unsigned)cdiv(nslices, 8), (unsigned)batch), dim3(256), lds, ctx->stream, D, sc, npad,
                     (int)klen, nslices, Ppad, plane, d_tiles, nwg, partial, nbuf == 2 ? 2 : 1, 0);
  VIPMI_CHECK_HIP(hipGetLastError());
  int rb = (int)cdiv((int64_t)nwg * 4096, 256);
  if (rb > 4096) rb = 4096;
  if (batch > 1 && rb > 64) rb = 64;
  hipLaunchKernelGGL(gram_i8_reduce_kernel, dim3((unsigned)rb, (unsigned)batch), dim3(256), 0, ctx->stream, partial, d_tiles, nwg,
                     nslices, (int)n, G);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

}  // namespace

// ---- the same Gram matrix one block of 64 rows at a time (rows arriving from the host: vipmi_pca_fullframe_hostin_f32) ----------
// G = M M^T needs every row, but tile (i, j) only the row blocks i and j: while the upload of block b + 1 is on the link, block b is
// split into its digit planes and multiplied with the blocks 0 .. b that are already there.  Same slices, same partial sums, same
// order of additions in the reduction as run(): the matrix is bit-identical to gram_i8_f32's (mode 1).  The tile list is ordered by
// the later block: tiles (0, b) .. (b, b) are workgroups b (b + 1) / 2 .. of the list.
int gram_i8_inc_begin(vipmi_ctx* ctx, int64_t n, int64_t P, int64_t ld, GramI8Inc* st) {
  constexpr int S = 5;
  st->n = n; st->P = P; st->ld = ld;
  st->npad = (int)cdiv(n, 64) * 64;
  st->nt = st->npad / 64;
  std::vector<int2> tiles;
  for (int j = 0; j < st->nt; ++j)
    for (int i = 0; i <= j; ++i) tiles.push_back(int2{i, j});
  st->nwg = (int)tiles.size();
  int64_t want = cdiv((int64_t)6 * ctx->num_cu, (int64_t)st->nwg);          // (the slice rule of run(): identical partial sums)
  if (ctx->opt("gram_i8_slices", 0) > 0) want = ctx->opt("gram_i8_slices", 0);
  if (want < 1) want = 1;
  int64_t klen = cdiv(cdiv(P, want), 64) * 64;
  if (klen < 256) klen = 256;
  if (klen > 8192) klen = 8192;
  st->klen = klen;
  st->nslices = (int)cdiv(P, klen);
  st->Ppad = (int64_t)st->nslices * klen;
  st->plane = (int64_t)st->npad * st->Ppad;
  VIPMI_TRY(ws(ctx, "gram_i8_digits", (size_t)S * st->plane, &st->D));
  VIPMI_TRY(ws(ctx, "gram_i8_scale", (size_t)st->npad * st->nslices, &st->sc));
  VIPMI_TRY(ws(ctx, "gram_i8_partial", (size_t)st->nslices * st->nwg * 4096, &st->partial));
  char key[64];
  snprintf(key, sizeof key, "i8inc/%d", st->nt);
  void* p = nullptr;
  VIPMI_TRY(ctx->upload_cached("gram_i8_tiles_inc", key, tiles.data(), sizeof(int2) * st->nwg, &p));
  st->d_tiles = reinterpret_cast<int2*>(p);
  return VIPMI_OK;
}

int gram_i8_inc_block(vipmi_ctx* ctx, const GramI8Inc& st, const float* M, int block) {
  constexpr int S = 5, KEEP = 1;
  StageScope sc(ctx, "gram");
  {
    const dim3 sg((unsigned)st.nslices, 64u, 1u);
    if (st.klen <= 4096)
      hipLaunchKernelGGL((gram_split_kernel<S, 1>), sg, dim3(256), 0, ctx->stream, M, (int)st.n, st.P, st.ld, (int)st.klen, st.nslices, st.Ppad,
                         st.plane, st.D, st.sc, (int64_t)0, st.npad, 64 * block);
    else
      hipLaunchKernelGGL((gram_split_kernel<S, 2>), sg, dim3(256), 0, ctx->stream, M, (int)st.n, st.P, st.ld, (int)st.klen, st.nslices, st.Ppad,
                         st.plane, st.D, st.sc, (int64_t)0, st.npad, 64 * block);
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  int nbuf = (int)ctx->opt("gram_i8_nbuf", 0);
  if (nbuf != 1 && nbuf != 2) nbuf = 2;
  const size_t lds = (size_t)(nbuf == 2 ? 2 : 1) * 2 * S * 64 * 64;
  const bool dma = nbuf == 2 && ctx->opt("gram_i8_dma", 1) != 0;
  auto kern = dma ? gram_i8_kernel<S, KEEP, true> : gram_i8_kernel<S, KEEP, false>;
  VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds));
  const int wg0 = block * (block + 1) / 2, cnt = block + 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * cnt), (unsigned)cdiv(st.nslices, 8), 1u), dim3(256), lds, ctx->stream, st.D, st.sc, st.npad,
                     (int)st.klen, st.nslices, st.Ppad, st.plane, st.d_tiles, st.nwg, st.partial, nbuf == 2 ? 2 : 1, wg0);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

int gram_i8_inc_end(vipmi_ctx* ctx, const GramI8Inc& st, double* G) {
  StageScope sc(ctx, "gram");
  int rb = (int)cdiv((int64_t)st.nwg * 4096, 256);
  if (rb > 4096) rb = 4096;
  hipLaunchKernelGGL(gram_i8_reduce_kernel, dim3((unsigned)rb, 1u), dim3(256), 0, ctx->stream, st.partial, st.d_tiles, st.nwg, st.nslices,
                     (int)st.n, G);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

// ---- ragged batch: the Gram matrices of the SEGMENTS of one matrix (annular PCA: every annulus segment a column range) --------
// M[n][Ptot] holds the segment matrices side by side, every segment a whole number of K-slices of klen columns (zero padded);
// G_all[s] = M[:, seg s] M[:, seg s]^T.  ONE split launch and ONE product launch over all slices -- the partial tiles of a slice
// do not care whose segment the slice belongs to -- and a reduction that sums every segment's own slice range
// (seg_slice[s] .. seg_slice[s + 1], device array).  Eight annuli of C3: 8 float64-MFMA Gram launches + 8 reductions
// (1.43 ms) become three launches.
__global__ void gram_i8_reduce_ragged_kernel(const double* __restrict__ partial, const int2* __restrict__ wgtiles, int nwg,
                                             const int32_t* __restrict__ seg_slice, int n, double* __restrict__ G) {
  const int64_t total = (int64_t)nwg * 4096;
  const int s0 = seg_slice[blockIdx.y], s1 = seg_slice[blockIdx.y + 1];
  G += (int64_t)blockIdx.y * n * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int wg = (int)(e >> 12), wave = (int)(e >> 10) & 3, rem = (int)(e & 1023);
    const int blk = rem >> 8, idx = rem & 255;
    const int bi = blk >> 1, bj = blk & 1, wi = wave >> 1, wj = wave & 1;
    const int2 t = wgtiles[wg];
    if (t.x == t.y && (wi > wj || (wi == wj && bj < bi))) continue;          // never written: lower part of a diagonal tile
    const int gi = t.x * 64 + wi * 32 + bi * 16 + (idx >> 4), gj = t.y * 64 + wj * 32 + bj * 16 + (idx & 15);
    if (gi >= n || gj >= n) continue;
    double s = 0.0;                         // (the slices in their order: deterministic)
    int sl = s0;
    for (; sl + 8 <= s1; sl += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(sl + u) * total + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sl < s1; ++sl) s += partial[(int64_t)sl * total + e];
    G[(int64_t)gi * n + gj] = s;
    G[(int64_t)gj * n + gi] = s;
  }
}

int gram_i8_ragged_f32(vipmi_ctx* ctx, const float* M, int64_t n, int64_t Ptot, int64_t klen, const int32_t* seg_slice,
                       int64_t nseg, double* G_all) {
  constexpr int S = 5, KEEP = 1;
  VIPMI_REQUIRE(M && seg_slice && G_all && n > 0 && nseg > 0, "gram_i8_ragged: bad arguments");
  VIPMI_REQUIRE(klen >= 256 && klen <= 4096 && (klen & 63) == 0 && Ptot > 0 && Ptot % klen == 0, "gram_i8_ragged: bad slice length");
  VIPMI_REQUIRE((reinterpret_cast<uintptr_t>(M) & 15) == 0 && nseg <= 65535, "gram_i8_ragged: unaligned input / too many segments");
  StageScope scope(ctx, "gram");
  const int npad = (int)cdiv(n, 64) * 64, nt = npad / 64;
  std::vector<int2> tiles;
  if (nt <= 8) {
    for (int i = 0; i < nt; ++i)
      for (int j = i + 1; j < nt; ++j) tiles.push_back(int2{i, j});
    for (int i = 0; i < nt; ++i) tiles.push_back(int2{i, i});
  } else {
    for (int I = 0; I < nt; I += 8)
      for (int J = I; J < nt; J += 8)
        for (int i = I; i < std::min(I + 8, nt); ++i)
          for (int j = std::max(J, i); j < std::min(J + 8, nt); ++j) tiles.push_back(int2{i, j});
  }
  const int nwg = (int)tiles.size();
  const int nslices = (int)(Ptot / klen);
  const int64_t plane = (int64_t)npad * Ptot;
  int8_t* D = nullptr;
  double *sc = nullptr, *partial = nullptr;
  VIPMI_TRY(ws(ctx, "gram_i8_digits", (size_t)S * plane, &D));
  VIPMI_TRY(ws(ctx, "gram_i8_scale", (size_t)npad * nslices, &sc));
  VIPMI_TRY(ws(ctx, "gram_i8_partial", (size_t)nslices * nwg * 4096, &partial));
  int2* d_tiles = nullptr;
  {
    char key[64];
    snprintf(key, sizeof key, "i8/%d", nt);                 // (the tile list of run(): same key, same table)
    void* p = nullptr;
    VIPMI_TRY(ctx->upload_cached("gram_i8_tiles", key, tiles.data(), sizeof(int2) * nwg, &p));
    d_tiles = reinterpret_cast<int2*>(p);
  }
  hipLaunchKernelGGL((gram_split_kernel<S, 1>), dim3((unsigned)nslices, (unsigned)npad, 1u), dim3(256), 0, ctx->stream, M, (int)n, Ptot, Ptot,
                     (int)klen, nslices, Ptot, plane, D, sc, (int64_t)0, npad, 0);
  VIPMI_CHECK_HIP(hipGetLastError());
  const size_t lds = (size_t)2 * 2 * S * 64 * 64;
  const bool dma = ctx->opt("gram_i8_dma", 1) != 0;
  auto kern = dma ? gram_i8_kernel<S, KEEP, true> : gram_i8_kernel<S, KEEP, false>;
  VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * nwg), (unsigned)cdiv(nslices, 8), 1u), dim3(256), lds, ctx->stream, D, sc, npad, (int)klen,
                     nslices, Ptot, plane, d_tiles, nwg, partial, 2, 0);
  VIPMI_CHECK_HIP(hipGetLastError());
  int rb = (int)cdiv((int64_t)nwg * 4096, 256);
  if (rb > 128) rb = 128;
  hipLaunchKernelGGL(gram_i8_reduce_ragged_kernel, dim3((unsigned)rb, (unsigned)nseg), dim3(256), 0, ctx->stream, partial, d_tiles, nwg,
                     seg_slice, (int)n, G_all);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

// where gram_f32 takes the int8 path by itself (the rule of gram.hip: measured to pay from 256 rows and 2^25 elements)
bool gram_i8_default_path(vipmi_ctx* ctx, int64_t n, int64_t P) {
  return ctx->opt("gram_i8", -1) < 0 && ctx->opt("gram_f32", 0) == 0 && n >= 256 && P >= 32768 && n * P >= ((int64_t)1 << 25);
}

// G[batch][n][n] = M M^T for `batch` float32 matrices [n][P] (row length ld; problems n * ld apart) on the int8 matrix cores.
// mode 1: 5 digits, 19 products (7e-12 max|G|); mode 2: 6 digits, 26 products (2e-15).
int gram_i8_f32(vipmi_ctx* ctx, const float* M, int64_t n, int64_t P, int64_t ld, double* G, int64_t batch, int mode) {
  VIPMI_REQUIRE((reinterpret_cast<uintptr_t>(M) & 3) == 0, "gram_i8: unaligned input");
  if (mode >= 2) return run<6, 1>(ctx, M, n, P, ld, G, batch);
  return run<5, 1>(ctx, M, n, P, ld, G, batch);
}

}  // namespace vipmi
