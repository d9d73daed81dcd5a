// derotate_direct2.hip -- "real-split" DIRECT derotation: the generic path for padded lengths that are not a power of
// two (odd frame sizes -- the usual case in practice: 101 px -> Le = 402 = 2*3*67, 201 px -> Le = 802 = 2*401 --
// and even sizes such as 200 px -> Le = 800).
//
// Same decomposition as derotate_fft2.hip (three passes of REAL circular sinc shifts plus rank-one Nyquist
// corrections, formulas in that file's header; re-validated against the reference for odd and even frame sizes and
// all quadrants to 1e-15 in float64), but every real shift is evaluated as the correlation it is,
//     (R_s x)[m] = sum_j x[j] r(m - j - s),      r(t) = sin(pi t) cos(pi t / Le) / (Le sin(pi t / Le))
// using the zero structure of the problem (N of Le inputs or outputs are live in every pass): 12 N^3 real
// multiply-adds per frame, against 44 N^3 for the complex-field correlation of derotate.hip, and with
//   * one LINEAR table of kernel values per line in LDS, T[u] = r(u + shift) for u in [-nin, nout): since
//     sin(pi (d - s)) = -(-1)^d sin(pi s), an entry costs one float32 sincos instead of three float64 ones;
//   * a register-blocked Toeplitz product: a lane owns 4 consecutive outputs, inputs are consumed 4 at a time, so one
//     step is 2 aligned 16-byte LDS reads (4 inputs broadcast + 4 new table entries; the other 4 table entries are
//     the previous step's) for 16 FMAs;
//   * the column pass works on 16-column tiles staged through LDS (64-byte row segments).
// K = sum_X R_{s_X} beta (the one term that couples all columns) is evaluated in the frequency domain with explicit
// DFT sums over an LDS table of the Le-th roots of unity (8 N^2 complex multiply-adds per frame).
#include "common.h"
#include "rot_common.h"
#include "fft_wave.h"

namespace vipmi {

namespace {

struct AuxD {          // per-batch auxiliary arrays (device), as in derotate_fft2.hip
  float* beta;         // [nf][N]
  float* bf;           // [nf]
  float* kv;           // [nf][N]
  float* gam;          // [nf][Le]   (-1)^X gamma_X
  float* gsum;         // [nf]
  const float* cotab;  // [Le] cot(pi m/Le)
};

__device__ __forceinline__ float sin_pi_d(double s) {     // sin(pi s), argument reduced in float64
  // reduce to [-1/2, 1/2] (NOT [-1, 1]: float32 would lose the distance to the zero at +-1, and the table divides
  // this by an equally small sine, so the relative error matters): sin(pi s) = (-1)^n sin(pi (s - n))
  const double n = rint(s);
  const float v = sinpif((float)(s - n));
  return (((long long)n) & 1) ? -v : v;
}

// LDS position of kernel-table entry idx: one 16-byte pad after every 16 entries.  The lanes of the Toeplitz product
// read 16-byte table blocks 64 bytes apart (a lane pair owns 16 consecutive outputs); unpadded, those addresses fall
// on 4 of the 16 bank groups (8-way conflict), padded (80 bytes apart) they cover all of them (2-way, the minimum
// for 32 distinct 16-byte blocks).  Aligned blocks of 4 entries stay contiguous.
__device__ __host__ __forceinline__ int tsw(int idx) { return idx + ((idx >> 4) << 2); }

// cotab[m] = cot(pi m / Le), m = 1 .. Le-1 (cotab[0] unused): filled once per call in float64 arithmetic.
__global__ void ds_cotab_kernel(float* __restrict__ cotab, int Le) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= Le) return;
  double sn, cs;
  sincospi((double)m / (double)Le, &sn, &cs);
  cotab[m] = (m == 0) ? 0.f : (float)(cs / sn);
}

// T[u + OFF] = r(u + d0 - s) for u in [-OFF, nout_pad): d0 = (first output index) - (first input index).
// r(t) = sin(pi t) cos(pi t/Le) / (Le sin(pi t/Le)) at t = d - s, d integer.  With s = n + f (n = rint(s), |f| <= 1/2)
// and m = d - n:  sin(pi (d - s)) = -(-1)^m sin(pi f)  and  cot(pi (m - f)/Le) = cot(alpha + beta), alpha = pi m/Le on
// the grid of `cotab`, beta = -pi f/Le tiny, so an entry is
//     r = -(-1)^m sin(pi f)/Le * (cot(alpha) - tan(beta)) / (1 + cot(alpha) tan(beta))        (m != 0 mod Le)
//     r = -sin(pi f)/(Le tan(beta))  (m = 0 mod Le: the pole, -> 1 as f -> 0)
// -- two FMAs and one reciprocal per entry, the two transcendental values once per line (an entry used to cost a
// float64 argument reduction, a sincos and a full division: as much as the Toeplitz product itself at 300 px).
// No cancellation: |alpha| >= pi/Le > 2 |beta| away from the pole; where cot(alpha) ~ tan(beta) the entry is ~ 0.
__device__ __forceinline__ void fill_table(float* __restrict__ T, int OFF, int nout_pad, int d0, double s, int Le,
                                           int tid, int nthreads, const float* __restrict__ cotab) {
  const double ns = rint(s);
  const double f = s - ns;                           // exact
  const float sf = sinpif((float)f);
  float tsn, tcs;
  sincospif((float)(f / (double)Le), &tsn, &tcs);
  const float tb = -tsn / tcs;                       // tan(beta)
  const float scale = sf / (float)Le;
  const bool integer_shift = (f == 0.0);
  const int total = OFF + nout_pad;
  // m = d - n for entry e = tid:  (tid - OFF + d0 - n) mod Le, then advanced by nthreads per iteration
  const int m0 = tid - OFF + d0 - (int)ns;           // |m0| < 2 Le for every pass (|s| < Le/2, |d| < Le)
  int mm = m0;
  while (mm < 0) mm += Le;
  while (mm >= Le) mm -= Le;
  int par = m0 & 1;                                  // parity of m (two's complement: valid for negatives)
  int step = nthreads;
  while (step >= Le) step -= Le;
  const int pstep = nthreads & 1;
  for (int e = tid; e < total; e += nthreads) {
    float r;
    if (integer_shift) {
      r = (mm == 0) ? 1.0f : 0.0f;
    } else {
      float cotv;
      if (mm == 0) {
        cotv = 1.0f / tb;
      } else {
        const float ca = cotab[mm];
        const float den = fmaf(ca, tb, 1.0f);
        float rc = __builtin_amdgcn_rcpf(den);
        rc = rc * fmaf(-den, rc, 2.0f);               // one Newton step
        cotv = (ca - tb) * rc;
      }
      r = (par ? scale : -scale) * cotv;              // -(-1)^m
    }
    T[tsw(e)] = r;
    mm += step;
    if (mm >= Le) mm -= Le;
    par ^= pstep;
  }
}

// Register-blocked Toeplitz product of one line on packed FP32:  y[m] = sum_j x[j] T[m - j + OFF].
// A lane owns the TNA outputs m = M + 2a + p (a < TNA; M a multiple of 2 TNA, p the lane's parity class) and consumes the
// inputs in PAIRS, so that one v_pk_fma_f32 performs two multiply-adds whose operands are both aligned register pairs:
//   p = 1:  pairs (x[k], x[k+1]);      p = 0:  pairs (x[k-1], x[k]) = aligned pairs of the shifted copy xo[k] = x[k-1];
// in both cases the kernel values of output a and pair k are the aligned pair T[i], T[i+1], i = M + 2a - k + OFF, taken
// in swapped order (op_sel) -- the parity split is what keeps every table pair aligned (owning consecutive outputs,
// half of the pairs straddle two registers and cost a move each).  The lanes keep two partial sums per output.
// One step = 4 inputs x TNA outputs: 2 TNA packed FMAs for 2 aligned 16-byte LDS reads (4 inputs broadcast + the 4 new
// table entries; the others are the previous steps').  X: x (p = 1) or xo (p = 0), nin_pad floats, zero padded (at least one
// zero after the last input), 16-byte aligned; T: OFF == nin_pad, entries up to OFF + M + 2 TNA - 1 valid.
typedef float f2 __attribute__((ext_vector_type(2)));
// acc + (w.y x.x, w.x x.y): one packed FMA, the swapped operand in src0
__device__ __forceinline__ f2 fma_swz(f2 w, f2 x, f2 acc) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(x));
  return acc;
}
constexpr int TNA = 8;                   // outputs per lane: 4 inputs x 8 outputs = 16 packed FMAs per 2 LDS reads (with 4
                                         // outputs per lane the passes were bound by LDS bandwidth, not by the FMAs)
template <int NA>
__device__ __forceinline__ void toeplitz_par(const float* __restrict__ X, int nin_pad, const float* __restrict__ T,
                                             int OFF, int M, float (&out)[NA], int k_begin = 0, int k_end = -1) {
  if (k_end < 0) k_end = nin_pad;                    // [k_begin, k_end): multiples of 4 (partial sums over the inputs)
  f2 acc[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) acc[a] = f2{0.f, 0.f};
  int b = M + OFF - k_begin;                         // multiple of 4
  float4 w[NA / 2];                                 // w[i] = T[b + 4 i .. b + 4 i + 3]
#pragma unroll
  for (int i = 0; i < NA / 2; ++i) w[i] = *reinterpret_cast<const float4*>(T + tsw(b + 4 * i));
  for (int k = k_begin; k < k_end; k += 4) {
    const float4 wm = *reinterpret_cast<const float4*>(T + tsw(b - 4));
    const float4 xv = *reinterpret_cast<const float4*>(X + k);
    const f2 x0 = f2{xv.x, xv.y}, x1 = f2{xv.z, xv.w};
    // table pairs W(j) = (T[b + 2 j], T[b + 2 j + 1]), j = -1 .. NA - 1, multiplied in SWAPPED order: the swizzle sits on src0
    // (fma_swz) -- left to the compiler it lands on src1 (`op_sel:[0,1,0]`), the operand form gfx950 gets wrong beside
    // another wave's 16x16x64-i8 / 16x16x32-bf16 MFMA (common.h, VIPMI_NO_PK32)
    f2 W[NA + 1];
    W[0] = f2{wm.z, wm.w};
#pragma unroll
    for (int i = 0; i < NA / 2; ++i) {
      W[1 + 2 * i] = f2{w[i].x, w[i].y};
      W[2 + 2 * i] = f2{w[i].z, w[i].w};
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) acc[a] = fma_swz(W[a], x1, fma_swz(W[a + 1], x0, acc[a]));
#pragma unroll
    for (int i = NA / 2 - 1; i > 0; --i) w[i] = w[i - 1];
    w[0] = wm;
    b -= 4;
  }
#pragma unroll
  for (int a = 0; a < NA; ++a) out[a] = acc[a].x + acc[a].y;
}

__device__ __forceinline__ void src_map_d(int q, int Y, const RotGeom& g, int& base, int& stride) {
  switch (q) {
    case 1: stride = g.N; base = -g.off * g.N + (g.Lc - Y - g.off); break;
    case 2: stride = -1; base = (g.Lc - Y - g.off) * g.N + (g.Lc - g.off); break;
    case 3: stride = -g.N; base = (g.Lc - g.off) * g.N + (Y - g.off); break;
    default: stride = 1; base = (Y - g.off) * g.N - g.off; break;
  }
}

__device__ __forceinline__ float block_sum(float v, float* red) {      // sum over the workgroup (<= 16 waves)
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < nw; ++w) t += red[w];
  __syncthreads();
  return t;
}

// ---- shear 1: one workgroup per data row; N inputs -> Le outputs ----
__global__ __launch_bounds__(1024) void ds_shear1(const float* __restrict__ in, const RotFrame* __restrict__ fr,
                                                  RotGeom g, float* __restrict__ A1r, AuxD aux, int f0, int Npad,
                                                  int Lpad) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* x = smem;                       // [Npad]
  float* xo = x + Npad;                  // [Npad + 8]  xo[k] = x[k-1]
  float* T = xo + Npad + 8;              // [tsw(Npad + Lpad)]
  float* red = T + tsw(Npad + Lpad);     // [16]
  const int fl = blockIdx.y, f = f0 + fl, yrel = blockIdx.x;
  const RotFrame p = fr[f];
  const int r0 = (p.q == 1 || p.q == 2) ? g.alt0 : g.off;
  const int c0 = (p.q == 2 || p.q == 3) ? g.alt0 : g.off;
  const int Y = r0 + yrel;
  const double s = p.a * (double)(Y - g.c);
  int b, st;
  src_map_d(p.q, Y, g, b, st);
  const float* frame = in + (int64_t)f * g.N * g.N;
  float alt = 0.f;
  for (int j = threadIdx.x; j < Npad; j += blockDim.x) {
    float v = 0.f;
    if (j < g.N) {
      const float t = frame[b + (c0 + j) * st];
      v = (t == t) ? t : 0.f;
    }
    x[j] = v;
    xo[j + 1] = v;
    alt += ((c0 + j) & 1) ? -v : v;
  }
  if (threadIdx.x == 0) xo[0] = 0.f;
  // outputs X = 0 .. Le-1, inputs at canvas columns c0 + j: d = X - (c0 + j) -> d0 = -c0
  fill_table(T, Npad, Lpad, -c0, s, g.Le, threadIdx.x, blockDim.x, aux.cotab);
  alt = block_sum(alt, red);            // contains the barrier that publishes x and T
  float* orow = A1r + ((int64_t)fl * g.N + yrel) * g.Le;
  for (int t = threadIdx.x; 2 * TNA * (t >> 1) < g.Le; t += blockDim.x) {
    const int M = 2 * TNA * (t >> 1), par = t & 1;
    float acc[TNA];
    toeplitz_par(par ? x : xo, Npad, T, Npad, M, acc);
#pragma unroll
    for (int a = 0; a < TNA; ++a)
      if (M + 2 * a + par < g.Le) orow[M + 2 * a + par] = acc[a];
  }
  if (threadIdx.x == 0) aux.beta[fl * g.N + yrel] = sin_pi_d(s) * alt / (float)g.Le;
}

// Bf[f] = sum_Y (-1)^Y beta_Y
__global__ __launch_bounds__(256) void ds_bf(const RotFrame* __restrict__ fr, RotGeom g, AuxD aux, int f0) {
  __shared__ float red[16];
  const int fl = blockIdx.x;
  const RotFrame p = fr[f0 + fl];
  const int r0 = (p.q == 1 || p.q == 2) ? g.alt0 : g.off;
  float s = 0.f;
  for (int y = threadIdx.x; y < g.N; y += blockDim.x) {
    const float b = aux.beta[fl * g.N + y];
    s += ((r0 + y) & 1) ? -b : b;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) aux.bf[fl] = s;
}

// K[m] = Re sum_k Bhat[k] g[k] e^{2 pi i k (off + m)/Le},  Bhat[k] = sum_y beta[y] e^{-2 pi i k (r0 + y)/Le},
// g(k) = sum_X exp(-2 pi i ks b (X - c)/Le)/Le in closed form (ks = signed frequency; Nyquist: real part).
// Two launches over a grid (frames, S): phase 1 writes the spectrum H of a frame in S slices to global memory, phase 2
// computes S slices of K from the whole H.  (One workgroup per frame did both: 0.6 ms for a 2048-pixel frame, a third of
// the derotation of a handful of such frames.)
__global__ VIPMI_NO_PK32 __launch_bounds__(1024) void ds_aux_k(const RotFrame* __restrict__ fr, RotGeom g, AuxD aux, int f0, int phase,
                                                 float2* __restrict__ Hg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* root = reinterpret_cast<float2*>(smem);          // [Le] e^{+2 pi i t/Le}
  float2* H = root + g.Le;                                  // [Le]
  float* beta = reinterpret_cast<float*>(H + g.Le);         // [N]
  const int fl = blockIdx.x, S = gridDim.y, sl = blockIdx.y;
  const RotFrame p = fr[f0 + fl];
  const int r0 = (p.q == 1 || p.q == 2) ? g.alt0 : g.off;
  const int Le = g.Le;
  float2* Hf = Hg + (size_t)fl * Le;
  for (int t = threadIdx.x; t < Le; t += blockDim.x) {
    float sn, cs;
    sincospif(2.0f * (float)t / (float)Le, &sn, &cs);
    root[t] = make_float2(cs, sn);
  }
  if (phase == 1) {
    for (int y = threadIdx.x; y < g.N; y += blockDim.x) beta[y] = aux.beta[fl * g.N + y];
    __syncthreads();
    const int per = (Le + S - 1) / S, k0 = sl * per, k1 = (k0 + per < Le) ? k0 + per : Le;
    for (int k = k0 + threadIdx.x; k < k1; k += blockDim.x) {
      float re = 0.f, im = 0.f;
      int idx = (int)(((long long)k * r0) % Le);             // k (r0 + y) mod Le, advanced by k per step
      for (int y = 0; y < g.N; ++y) {
        const float2 w = root[idx];
        re += beta[y] * w.x;                                  // e^{-i phi} = (cos, -sin)
        im -= beta[y] * w.y;
        idx += k;
        if (idx >= Le) idx -= Le;
      }
      const int ks = (k < Le / 2) ? k : k - Le;
      const double kbv = (double)ks * p.b;
      const double den = sinpi(kbv / (double)Le);
      const double ratio = (fabs(den) < 1e-300) ? (double)Le : sinpi(kbv) / den;
      double sn, cs;
      sincospi(-2.0 * kbv * (0.5 * (double)(Le - 1) - (double)g.c) / (double)Le, &sn, &cs);
      float gre = (float)(ratio * cs / (double)Le), gim = (float)(ratio * sn / (double)Le);
      if (k == Le / 2) gim = 0.f;
      Hf[k] = make_float2(re * gre - im * gim, re * gim + im * gre);
    }
    return;
  }
  for (int k = threadIdx.x; k < Le; k += blockDim.x) H[k] = Hf[k];
  __syncthreads();
  const int per = (g.N + S - 1) / S, m0 = sl * per, m1 = (m0 + per < g.N) ? m0 + per : g.N;
  for (int m = m0 + threadIdx.x; m < m1; m += blockDim.x) {
    const int Yo = g.off + m;
    float re = 0.f;
    int idx = 0;
    for (int k = 0; k < Le; ++k) {
      const float2 w = root[idx], h = H[k];
      re += h.x * w.x - h.y * w.y;
      idx += Yo;
      if (idx >= Le) idx -= Le;
    }
    aux.kv[fl * g.N + m] = re;
  }
}

// ---- shear 2: one workgroup per 16-column tile; N live input rows -> N output rows (crop window) ----
template <int CT>          // columns per tile (16 = one 64-byte segment per row; fewer for large frames: LDS)
__global__ __launch_bounds__(1024) void ds_shear2(const float* __restrict__ A1r, const RotFrame* __restrict__ fr,
                                                  RotGeom g, float* __restrict__ A2r, AuxD aux, int f0, int Npad) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int CS = Npad + 8;                           // column stride in LDS (16-byte aligned, spreads the banks)
  float* xin = smem;                                 // [CT][CS]   column-major copy of the tile
  float* xsh = xin + CT * CS;                        // [CT][CS]   the same shifted by one row: xsh[k] = xin[k-1]
  const int TS = tsw(2 * Npad);                      // table stride per column
  float* T = xsh + CT * CS;                          // [CT][TS]
  float* yout = T + CT * TS;                         // [CT][CS]
  float* alts = yout + CT * CS;                      // [CT]
  float* sps = alts + CT;                            // [CT]
  const int fl = blockIdx.y, f = f0 + fl, X0 = blockIdx.x * CT;
  const RotFrame p = fr[f];
  const int r0 = (p.q == 1 || p.q == 2) ? g.alt0 : g.off;
  const int ncol = (g.Le - X0 < CT) ? (g.Le - X0) : CT;
  const float* src = A1r + (int64_t)fl * g.N * g.Le + X0;
  // tile in: 64-byte row segments, transposed into LDS
  for (int e = threadIdx.x; e < Npad * CT; e += blockDim.x) {
    const int row = e / CT, c = e % CT;
    const float v = (row < g.N && c < ncol) ? src[(int64_t)row * g.Le + c] : 0.f;
    xin[c * CS + row] = v;
    xsh[c * CS + row + 1] = v;
  }
  if (threadIdx.x < CT) xsh[threadIdx.x * CS] = 0.f;
  // kernel tables: outputs at canvas rows off + m, inputs at rows r0 + y: d0 = off - r0
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  for (int c = wave; c < CT; c += nw) {
    const double s = p.b * (double)(X0 + c - g.c);
    fill_table(T + c * TS, Npad, Npad, g.off - r0, s, g.Le, lane, 64, aux.cotab);
    if (lane == 0) sps[c] = sin_pi_d(s) / (float)g.Le;
  }
  __syncthreads();
  // alternating sums of the columns (gamma) -- one wave per column
  for (int c = wave; c < CT; c += nw) {
    float a = 0.f;
    for (int y = lane; y < g.N; y += 64) {
      const float v = xin[c * CS + y];
      a += ((r0 + y) & 1) ? -v : v;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
    if (lane == 0) alts[c] = a;
  }
  // Toeplitz products: tasks = (column, block of 4 outputs)
  const int nblk = Npad / TNA;
  for (int t = threadIdx.x; t < CT * nblk; t += blockDim.x) {
    const int c = t / nblk, tt = t % nblk;
    const int M = 2 * TNA * (tt >> 1), par = tt & 1;
    float acc[TNA];
    toeplitz_par((par ? xin : xsh) + c * CS, Npad, T + c * TS, Npad, M, acc);
#pragma unroll
    for (int a = 0; a < TNA; ++a) yout[c * CS + M + 2 * a + par] = acc[a];
  }
  __syncthreads();
  // tile out with the rank-one correction  - sin(pi s_X) (-1)^X Bf/Le (-1)^Y
  const float bfl = aux.bf[fl];
  float* dst = A2r + (int64_t)fl * g.N * g.Le + X0;
  for (int e = threadIdx.x; e < g.N * CT; e += blockDim.x) {
    const int m = e / CT, c = e % CT;
    if (c < ncol) {
      const int X = X0 + c;
      const float kc = sps[c] * ((X & 1) ? -bfl : bfl);
      const float sg = ((g.off + m) & 1) ? -1.f : 1.f;
      dst[(int64_t)m * g.Le + c] = yout[c * CS + m] - sg * kc;
    }
  }
  if (threadIdx.x < ncol) {
    const int X = X0 + threadIdx.x;
    const float sn = sps[threadIdx.x];
    aux.gam[fl * g.Le + X] = ((X & 1) ? -sn : sn) * alts[threadIdx.x];
  }
}

// Gam[f] = sum_X (-1)^X gamma_X
__global__ __launch_bounds__(256) void ds_gamma(AuxD aux, int Le) {
  __shared__ float red[16];
  const int fl = blockIdx.x;
  float s = 0.f;
  for (int x = threadIdx.x; x < Le; x += blockDim.x) s += aux.gam[fl * Le + x];
  s = block_sum(s, red);
  if (threadIdx.x == 0) aux.gsum[fl] = s;
}

// ---- shear 3: one workgroup per output row; Le inputs -> N outputs, corrections, mask restore ----
template <int NA>           // outputs per lane: 8 for long rows, 4 where N / 8 lanes would leave most of a wave idle
__global__ __launch_bounds__(1024) void ds_shear3(const float* __restrict__ A2r, const RotFrame* __restrict__ fr,
                                                 RotGeom g, const float* __restrict__ in, float* __restrict__ out,
                                                 AuxD aux, int f0, int mask_nan, int mask_zero, float mask_v, int Npad, int Lpad) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* x = smem;                       // [Lpad]
  float* xo = x + Lpad;                  // [Lpad + 8]  xo[k] = x[k-1]
  float* T = xo + Lpad + 8;              // [tsw(Lpad + Npad)]
  float* part = T + tsw(Lpad + Npad);    // [groups][Npad] partial sums
  const int fl = blockIdx.y, f = f0 + fl, m = blockIdx.x;
  const RotFrame p = fr[f];
  const int Y = g.off + m;
  const double s = p.a * (double)(Y - g.c);
  const float* irow = A2r + ((int64_t)fl * g.N + m) * g.Le;
  for (int j = threadIdx.x; j < Lpad; j += blockDim.x) {
    const float v = (j < g.Le) ? irow[j] : 0.f;
    x[j] = v;
    xo[j + 1] = v;
  }
  if (threadIdx.x == 0) xo[0] = 0.f;
  // outputs at canvas columns off + jo, inputs at columns 0..Le-1: d0 = off
  fill_table(T, Lpad, Npad, g.off, s, g.Le, threadIdx.x, blockDim.x, aux.cotab);
  __syncthreads();
  const float gs = aux.gsum[fl];
  const float c1 = sin_pi_d(s) / (float)g.Le * (aux.kv[fl * g.N + m] + ((Y & 1) ? -gs : gs));
  const int64_t ob = ((int64_t)f * g.N + m) * g.N;
  // Le inputs -> only N outputs: a row has work for Npad / NA lanes (one wave at 511 px), so the INPUTS are split over
  // the lane groups of the workgroup (each group covers every output for its share of the inputs) and the partial
  // sums are added through LDS in a fixed order
  const int lg = Npad / NA;                                  // lanes that cover the outputs (Npad: multiple of 16)
  const int lgp = (lg + 63) / 64 * 64;                        // ... padded to whole waves
  const int ngrp = blockDim.x / lgp > 0 ? blockDim.x / lgp : 1;
  const int grp = threadIdx.x / lgp, tl = threadIdx.x % lgp;
  const int kq = (Lpad / 4 + ngrp - 1) / ngrp * 4;            // inputs per group (multiple of 4)
  if (grp < ngrp && tl < lg) {
    const int M = 2 * NA * (tl >> 1), par = tl & 1;
    const int k0 = (grp * kq < Lpad) ? grp * kq : Lpad, k1 = (k0 + kq < Lpad) ? k0 + kq : Lpad;   // may be empty
    float acc[NA];
    toeplitz_par(par ? x : xo, Lpad, T, Lpad, M, acc, k0, k1);
#pragma unroll
    for (int o = 0; o < NA; ++o) part[grp * Npad + M + 2 * o + par] = acc[o];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < g.N; j += blockDim.x) {
    float re = 0.f;
    for (int w = 0; w < ngrp; ++w) re += part[w * Npad + j];
    const float sg = ((g.off + j) & 1) ? -1.f : 1.f;
    re -= sg * c1;
    const float srcv = in[ob + j];
    if (mask_nan && !(srcv == srcv)) re = __uint_as_float(0x7fc00000u);
    if (mask_zero && srcv == mask_v) re = mask_v;
    out[ob + j] = re;
  }
}

// K vectors of a batch of frames: the frames times S slices of workgroups (few large frames would leave the chip empty)
static void launch_aux_k(vipmi_ctx* ctx, unsigned nf, size_t ldsk, const RotFrame* d_frames, const RotGeom& g, const AuxD& aux,
                         int f0, float2* Hg) {
  int S = nf > 0 ? (int)(ctx->num_cu / nf) : 1;
  if (S < 1) S = 1;
  if (S > 8) S = 8;
  hipLaunchKernelGGL(ds_aux_k, dim3(nf, S), dim3(1024), ldsk, ctx->stream, d_frames, g, aux, f0, 1, Hg);
  hipLaunchKernelGGL(ds_aux_k, dim3(nf, S), dim3(1024), ldsk, ctx->stream, d_frames, g, aux, f0, 2, Hg);
}

#include "derotate_conv.inc"

}  // namespace

int derotate_direct2(vipmi_ctx* ctx, const float* in, const RotFrame* d_frames, const RotGeom& g, int64_t n, float* out,
                     int mask_nan, int mask_zero, float mask_v) {
  // paddings: multiples of 2 TNA (a lane pair owns 2 TNA outputs) with at least one zero after the last input
  const int Npad = (int)cdiv(g.N + 1, 2 * TNA) * 2 * TNA, Lpad = (int)cdiv(g.Le + 1, 2 * TNA) * 2 * TNA;
  // 129 .. 512 px: the three passes as power-of-two circular convolutions (derotate_conv.inc) on blocked intermediates
  // (N, Le rounded up to multiples of 128); option rot_conv = 0 keeps the direct correlations
  const bool conv = ctx->opt("rot_conv", 1) != 0 && g.N > 128 && g.N <= 2048;       // (from 513 px: up to four parts per line)
  CvLayout lay;
  lay.nbr = (int)cdiv(g.N, 128) * 64;
  lay.nbc = (int)cdiv(g.Le, 128) * 64;
  const int64_t per_frame = conv ? (int64_t)lay.nbr * lay.nbc * 4 : (int64_t)g.N * g.Le;
  int64_t chunk = ctx->opt("rot_batch", 0);
  if (chunk <= 0) {
    const int64_t budget = ctx->opt("rot_ws_mb", 4096) * (int64_t)(1 << 20);
    chunk = budget / (2 * per_frame * (int64_t)sizeof(float));
  }
  if (chunk < 1) chunk = 1;
  if (chunk > n) chunk = n;
  if (chunk > 65535) chunk = 65535;
  float *A1r = nullptr, *A2r = nullptr;
  VIPMI_TRY(ws(ctx, "rot_a1", (size_t)(chunk * per_frame), &A1r));
  VIPMI_TRY(ws(ctx, "rot_a2", (size_t)(chunk * per_frame), &A2r));
  AuxD aux;
  VIPMI_TRY(ws(ctx, "rot_beta", (size_t)(chunk * g.N), &aux.beta));
  VIPMI_TRY(ws(ctx, "rot_bf", (size_t)chunk, &aux.bf));
  VIPMI_TRY(ws(ctx, "rot_kv", (size_t)(chunk * g.N), &aux.kv));
  VIPMI_TRY(ws(ctx, "rot_gam", (size_t)(chunk * g.Le), &aux.gam));
  VIPMI_TRY(ws(ctx, "rot_gsum", (size_t)chunk, &aux.gsum));
  float2* aux_H = nullptr;
  VIPMI_TRY(ws(ctx, "rot_auxH", (size_t)(chunk * g.Le), &aux_H));
  float* cotab = nullptr;
  VIPMI_TRY(ws(ctx, "rot_cotab", (size_t)g.Le, &cotab));
  hipLaunchKernelGGL(ds_cotab_kernel, dim3((unsigned)cdiv(g.Le, 256)), dim3(256), 0, ctx->stream, cotab, g.Le);
  aux.cotab = cotab;
  const size_t lds1 = (size_t)(2 * Npad + 8 + tsw(Npad + Lpad) + 16) * sizeof(float);
  const size_t ldsk = (size_t)(4 * g.Le + g.N) * sizeof(float);
  int CT = 16;
  auto lds2_for = [&](int ct) { return (size_t)(ct * (3 * Npad + 24 + tsw(2 * Npad)) + 2 * ct) * sizeof(float); };
  while (CT > 2 && lds2_for(CT) > 150 * 1024) CT >>= 1;
  const size_t lds2 = lds2_for(CT);
  const int na3 = g.N >= 256 ? 8 : 4;                          // outputs per lane in shear 3
  const int lg3 = (int)cdiv(Npad / na3, 64) * 64;             // lanes covering the outputs of one row in shear 3
  const int t3 = lg3 >= 256 ? lg3 : 256 / lg3 * lg3;          // lane groups splitting the inputs (256 threads)
  const size_t lds3 = (size_t)(2 * Lpad + 8 + tsw(Lpad + Npad) + (t3 / lg3) * Npad) * sizeof(float);
  VIPMI_REQUIRE(lds1 <= 160 * 1024 && ldsk <= 160 * 1024 && lds2 <= 160 * 1024 && lds3 <= 160 * 1024,
                "derotate(direct): frame size %d too large for the LDS-resident tables", g.N);
  VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(ds_shear1), (int)lds1));
  VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(ds_aux_k), (int)ldsk));
  const void* k2 = CT == 16  ? reinterpret_cast<const void*>(ds_shear2<16>)
                   : CT == 8 ? reinterpret_cast<const void*>(ds_shear2<8>)
                   : CT == 4 ? reinterpret_cast<const void*>(ds_shear2<4>)
                             : reinterpret_cast<const void*>(ds_shear2<2>);
  VIPMI_CHECK_HIP(set_dyn_lds(k2, (int)lds2));
  const void* k3 = na3 == 8 ? reinterpret_cast<const void*>(ds_shear3<8>) : reinterpret_cast<const void*>(ds_shear3<4>);
  VIPMI_CHECK_HIP(set_dyn_lds(k3, (int)lds3));
  // threads: one lane per TNA outputs
  auto threads_for = [](int nout, int cap) {
    int t = (int)cdiv(cdiv(nout, TNA), 64) * 64;
    return t < 64 ? 64 : (t > cap ? cap : t);
  };
  const int t1 = threads_for(g.Le, 1024);
  int t2 = (int)cdiv((int64_t)CT * (Npad / TNA), 64) * 64;
  if (t2 > 1024) t2 = 1024;
  if (t2 < 256) t2 = 256;
  for (int64_t f0 = 0; f0 < n; f0 += chunk) {
    const unsigned nf = (unsigned)((n - f0) < chunk ? (n - f0) : chunk);
    if (conv) {
      if (2 * g.N - 1 <= 512)
        VIPMI_TRY((conv_passes<fftw::Plan512>(ctx, in, d_frames, g, reinterpret_cast<float4*>(A1r), reinterpret_cast<float4*>(A2r), lay, aux, f0, (int)nf, out, mask_nan, mask_zero, mask_v, ldsk, aux_H)));
      else
        VIPMI_TRY((conv_passes<fftw::Plan1024>(ctx, in, d_frames, g, reinterpret_cast<float4*>(A1r), reinterpret_cast<float4*>(A2r), lay, aux, f0, (int)nf, out, mask_nan, mask_zero, mask_v, ldsk, aux_H)));
      continue;
    }
    ctx->tic("k_rot_s1");
    hipLaunchKernelGGL(ds_shear1, dim3(g.N, nf), dim3(t1), lds1, ctx->stream, in, d_frames, g, A1r, aux, (int)f0, Npad,
                       Lpad);
    ctx->toc("k_rot_s1");
    ctx->tic("k_rot_aux");
    hipLaunchKernelGGL(ds_bf, dim3(nf), dim3(256), 0, ctx->stream, d_frames, g, aux, (int)f0);
    launch_aux_k(ctx, nf, ldsk, d_frames, g, aux, (int)f0, aux_H);
    ctx->toc("k_rot_aux");
    ctx->tic("k_rot_s2");
    {
      const dim3 grid2((unsigned)cdiv(g.Le, CT), nf);
      if (CT == 16)
        hipLaunchKernelGGL(ds_shear2<16>, grid2, dim3(t2), lds2, ctx->stream, A1r, d_frames, g, A2r, aux, (int)f0, Npad);
      else if (CT == 8)
        hipLaunchKernelGGL(ds_shear2<8>, grid2, dim3(t2), lds2, ctx->stream, A1r, d_frames, g, A2r, aux, (int)f0, Npad);
      else if (CT == 4)
        hipLaunchKernelGGL(ds_shear2<4>, grid2, dim3(t2), lds2, ctx->stream, A1r, d_frames, g, A2r, aux, (int)f0, Npad);
      else
        hipLaunchKernelGGL(ds_shear2<2>, grid2, dim3(t2), lds2, ctx->stream, A1r, d_frames, g, A2r, aux, (int)f0, Npad);
    }
    ctx->toc("k_rot_s2");
    hipLaunchKernelGGL(ds_gamma, dim3(nf), dim3(256), 0, ctx->stream, aux, g.Le);
    ctx->tic("k_rot_s3");
    if (na3 == 8)
      hipLaunchKernelGGL(ds_shear3<8>, dim3(g.N, nf), dim3(t3), lds3, ctx->stream, A2r, d_frames, g, in, out, aux,
                         (int)f0, mask_nan, mask_zero, mask_v, Npad, Lpad);
    else
      hipLaunchKernelGGL(ds_shear3<4>, dim3(g.N, nf), dim3(t3), lds3, ctx->stream, A2r, d_frames, g, in, out, aux,
                         (int)f0, mask_nan, mask_zero, mask_v, Npad, Lpad);
    ctx->toc("k_rot_s3");
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  return VIPMI_OK;
}

}  // namespace vipmi
