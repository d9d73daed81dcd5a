// derotate_fft2.hip -- "real-split" FFT derotation: the default fast path for power-of-two padded lengths.
//
// The reference carries a COMPLEX field through its three shears (preproc/derotation.py:611-620) although the
// image is real: the only source of an imaginary part is the Nyquist bin, whose phase exp(i pi s) is not +-1.
// For a real line x the shift S_s x = ifft(fft(x) exp(-2 pi i f s)) splits exactly into
//     S_s x = R_s x  +  i * n_s(x),      n_s(x)[m] = sin(pi s) * alt(x)/Le * (-1)^m,   alt(x) = sum_j (-1)^j x[j]
// with R_s the REAL shift (Nyquist phase replaced by cos(pi s)).  Pushing this through the three shears
// (linearity) leaves three passes of real shifts plus rank-one corrections:
//   shear 1:  A1r[Y,:] = R_{a(Y-c)} x_Y                    beta_Y  = sin(pi s) alt(x_Y)/Le
//   shear 2:  A2r[:,X] = R_{b(X-c)} A1r[:,X] - sin(pi s_X) (-1)^X Bf/Le (-1)^Y ,   Bf = sum_Y (-1)^Y beta_Y
//             gamma_X  = sin(pi s_X) alt(A1r[:,X])/Le ,     Gam = sum_X (-1)^X gamma_X
//   shear 3:  out[Y,:] = R_{a(Y-c)} A2r[Y,:] - sin(pi s_Y)/Le (K[Y] + Gam (-1)^Y) (-1)^m
//             K[Y] = sum_X (R_{b(X-c)} beta)[Y] = ifft( fft(beta) g )[Y],  g(k) = sum_X exp(-2 pi i f_k b (X-c))
//                                                                               (closed form: a Dirichlet ratio)
// (verified against the reference to 1e-15 in float64).  Real shifts come two at a time out of one complex
// transform: z = x1 + i x2, Z = fft(z), W[k] = Z[k] A[k] + conj(Z[-k]) B[k] with A,B = (p1 +- p2)/2, w = ifft(W) =
// y1 + i y2 -- one extra LDS exchange (mirror bin -k) instead of a second forward+inverse pair.  Net: half the
// transforms and half the intermediate bytes (float32 A1r/A2r) of the complex formulation in derotate_fft.hip.
#include "common.h"
#include "rot_common.h"
#include "fft_wave.h"

namespace vipmi {

namespace {

using namespace fftw;

// sin(pi s), cos(pi s) with the argument reduced in float64 (|s| reaches hundreds of pixels), float32 evaluation
__device__ __forceinline__ void sincos_pi(double s, float& sn, float& cs) {
  const double r = s - 2.0 * rint(0.5 * s);      // in [-1, 1]
  sincospif((float)r, &sn, &cs);
}

// v = x1 + i x2 (D1) -> y1 + i y2 (D1), real shifts by s1 / s2; alt1/alt2 = alternating sums of x1 / x2.
// LIVE4: only the inputs at the frame's canvas positions [3L/8, 5L/8) are set (pruned first butterflies);
// KEEP4: only those output positions are produced (pruned last butterflies).
// dph (optional): the per-frame table of rs_phase_delta for s2 - s1 -- the phases of the second line then follow from
// those of the first by three complex multiplies instead of four more sincos evaluations.
template <class P, bool LIVE4 = false, bool KEEP4 = false, bool DELTA = false>
__device__ __forceinline__ void pair_shift(cf (&v)[P::VL], const Twiddles<P>& tw, cf* __restrict__ lds, double s1,
                                           double s2, int lane, int sub, float& alt1, float& alt2,
                                           float& sin1, float& sin2,      // sinX = sin(pi sX)/L
                                           const cf* __restrict__ dph = nullptr) {
  cf d_lane = mkcf(1.f, 0.f), d_z = d_lane, d_w = d_lane, d_pi = d_lane;
  if constexpr (DELTA) {                                 // issued before the transform: the loads are back long before they are used
    d_lane = dph[lane];
    d_z = dph[64];
    d_w = dph[65];
    d_pi = dph[66];
  }
  constexpr int R1 = P::R1, R2 = P::R2, R3 = P::R3;
  fft_forward<P, LIVE4>(v, tw, lds, lane, sub);
  const int u0 = sub * P::U3L;
#pragma unroll
  for (int ul = 0; ul < P::U3L; ++ul) {
    const int wq = lane + 64 * (u0 + ul), k1 = wq / R2, ka = wq % R2;
#pragma unroll
    for (int kb = 0; kb < R3; ++kb) lds[k1 * P::T1 + ka * P::T2 + kb] = v[ul * R3 + kb];
  }
  xbar<P>(tw);
  {
    const cf zn = lds[R3 / 2];          // Z[Nyquist]: (k1, ka, kb) = (0, 0, R3/2)
    alt1 = zn.x;
    alt2 = zn.y;
  }
  ShearPhase<P> p1, p2;
  p1.init(s1, lane, sub);
  p1.pa = mkcf(0.5f * p1.pa.x, 0.5f * p1.pa.y);        // W = q1 (Z + conj Z~)/2 + q2 (Z - conj Z~)/2: halves folded in
  float sn1, cn1, sn2, cn2;
  sincos_pi(s1, sn1, cn1);
  if constexpr (DELTA) {
    p2.init_from(p1, d_lane, d_z, d_w);
    sn2 = sn1 * d_pi.x + cn1 * d_pi.y;                 // sin / cos of pi (s1 + delta)
    cn2 = cn1 * d_pi.x - sn1 * d_pi.y;
  } else {
    p2.init(s2, lane, sub);
    p2.pa = mkcf(0.5f * p2.pa.x, 0.5f * p2.pa.y);
    sincos_pi(s2, sn2, cn2);
  }
  const float c1n = cn1 * (0.5f / (float)P::L), c2n = cn2 * (0.5f / (float)P::L);
  sin1 = sn1 * (1.0f / (float)P::L);
  sin2 = sn2 * (1.0f / (float)P::L);
#pragma unroll
  for (int ul = 0; ul < P::U3L; ++ul) {
    const int wq = lane + 64 * (u0 + ul), k1 = wq / R2, ka = wq % R2;
    // mirror bin -k: digit-wise complement, with the borrow chain ending at the first non-zero digit
    int k1m, kam;
    bool special = false;
    if (k1 > 0) {
      k1m = R1 - k1;
      kam = R2 - 1 - ka;
    } else if (ka > 0) {
      k1m = 0;
      kam = R2 - ka;
    } else {
      k1m = 0;
      kam = 0;
      special = true;
    }
    const int mbase = k1m * P::T1 + kam * P::T2;
#pragma unroll
    for (int kb = 0; kb < R3; ++kb) {
      const int kbm = special ? ((R3 - kb) % R3) : (R3 - 1 - kb);
      const cf zm = lds[mbase + kbm];
      cf q1 = cmul(p1.pa, p1.pb(kb)), q2 = cmul(p2.pa, p2.pb(kb));
      if (kb == R3 / 2 && special) {           // Nyquist bin: real multipliers cos(pi s)
        q1 = mkcf(c1n, 0.f);
        q2 = mkcf(c2n, 0.f);
      }
      // spectra of the two real lines: E = Z + conj(Z~) = 2 X1, O = Z - conj(Z~) = 2i X2;  W = q1 E + q2 O
      const cf z = v[ul * R3 + kb];
      const cf E = cadd_conj(z, zm), O = csub_conj(z, zm);
      v[ul * R3 + kb] = cmla(q2, O, cmul(q1, E));
    }
    p1.next_u();
    p2.next_u();
  }
  xbar<P>(tw);                               // every mirror read done before the inverse reuses the region
  fft_inverse<P, KEEP4>(v, tw, lds, lane, sub);
}

struct Aux {          // per-batch auxiliary arrays (device)
  float* beta;        // [nf][N]   beta_Y of the data rows
  float* bf;          // [nf]      Bf
  float* kv;          // [nf][N]   K[off + m]
  float* gam;         // [nf][L]   (-1)^X gamma_X
  float* gsum;        // [nf]      Gam
  cf* dph;            // [nf][2][DPH_STRIDE] phase factors of the 64-line increments (rs_phase_delta), blocked plans only
  // blocked A2r with the 2 x 2 blocks of row groups G and G + 1 side by side (option rot_pair_store, see rs_shear2_direct)
  int pair;
};

#define VIPMI_SLOT_PROLOGUE()                                                     \
  extern __shared__ __attribute__((aligned(16))) cf lds_all[];                    \
  const int lane = threadIdx.x & 63;                                              \
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);              \
  const int sub = wave % P::WPL, slot = wave / P::WPL;                            \
  cf* lds = lds_all + slot * P::LDS_ELEMS;                                        \
  Twiddles<P> tw;                                                                 \
  tw.init(twtab, lds_all + P::LPB * P::LDS_ELEMS, lane)

// Dynamic work distribution.  The shear kernels are launched with one persistent workgroup per CU, but when they
// share the GPU with another stream's eigensolver some of those workgroups only become resident once an earlier one
// retires; with a static grid-stride split such a launch takes twice as long.  Tasks are therefore handed out
// through per-XCD counters (8 words, zeroed before the launch): token t of XCD x is task (t/TPG*8 + x)*TPG + t%TPG, so
// TPG consecutive tasks stay on one XCD (= one L2).  The token of the NEXT task is requested before the transforms
// of the current one and consumed after them, which hides the atomic's round trip.
//   WPL == 1 (row kernels): every wave owns its line pair and fetches for itself -- no workgroup barrier;
//   otherwise / rs_shear2 : one token per workgroup, broadcast through two alternating LDS words.
template <class P, bool PER_WAVE>
struct Tasks {
  int* ctr;
  int xcd, pending, parity;
  int* sh;
  __device__ __forceinline__ void init(int* counters, cf* lds_all) {
    xcd = blockIdx.x & 7;
    ctr = counters + xcd * 32;        // one 128-byte line per queue
    sh = reinterpret_cast<int*>(lds_all + P::LPB * P::LDS_ELEMS + Twiddles<P>::PER_LANE * 64) + 8;
    parity = 0;
    pending = 0;
  }
  __device__ __forceinline__ void request() {
    pending = 0;
    if (PER_WAVE ? ((threadIdx.x & 63) == 0) : (threadIdx.x == 0)) pending = atomicAdd(ctr, 1);
  }
  template <int TPG>
  __device__ __forceinline__ int take() {       // contains a workgroup barrier unless PER_WAVE
    int t;
    if (PER_WAVE) {
      t = __builtin_amdgcn_readfirstlane(pending);
    } else {
      if (threadIdx.x == 0) sh[parity] = pending;
      __syncthreads();
      t = sh[parity];
      parity ^= 1;
    }
    return ((t / TPG) * 8 + xcd) * TPG + (t % TPG);
  }
};

// Workgroup-wide task sequence WITHOUT a workgroup barrier (one wave per line): all waves of a workgroup walk the same
// dynamic sequence of tasks (their line pairs are adjacent rows -- that is what makes the rot90-folded gather of
// shear 1 efficient), but each at its own pace, so that one wave's loads overlap the other waves' transforms instead of
// all eight waiting in the same phase.  The sequence lives in a 4-entry LDS ring: the first wave to need token i claims
// it (LDS compare-and-swap), fetches it from the XCD queue and publishes it; a wave may run at most 4 tokens ahead
// of the slowest one.  All waves see the same tokens, hence leave the loop at the same index.
// LDS words (the barrier counters of Twiddles, unused when WPL == 1): ring [0..3], ready [4], claim [5],
// consumed [8 + wave].
template <class P>
struct RingTasks {
  static_assert(P::WPL == 1 && P::WPB <= 8, "RingTasks: one wave per line, at most 8 waves");
  static constexpr int TPG = 8;
  int* ctr;
  volatile int* w;
  int xcd, i, wave;
  __device__ __forceinline__ void init(int* counters, cf* lds_all, int wave_) {
    xcd = blockIdx.x & 7;
    ctr = counters + xcd * 32;
    w = reinterpret_cast<volatile int*>(lds_all + P::LPB * P::LDS_ELEMS + Twiddles<P>::PER_LANE * 64);   // zeroed by tw.init
    i = 0;
    wave = wave_;
  }
  __device__ __forceinline__ int next() {
    int t = 0;
    if ((threadIdx.x & 63) == 0) {
      for (;;) {
        if (i < w[4]) {
          t = w[i & 3];
          break;
        }
        if (atomicCAS(const_cast<int*>(&w[5]), i, i + 1) == i) {
          if (i >= 4)
            for (int k = 0; k < P::WPB; ++k)
              while (w[8 + k] < i - 3) __builtin_amdgcn_s_sleep(1);
          t = atomicAdd(ctr, 1);
          w[i & 3] = t;
          __threadfence_block();
          w[4] = i + 1;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      w[8 + wave] = i + 1;
    }
    ++i;
    t = __builtin_amdgcn_readfirstlane(t);
    return ((t / TPG) * 8 + xcd) * TPG + (t % TPG);      // TPG consecutive tasks (adjacent rows) stay on one XCD / L2
  }
};

// ---- blocked intermediates (plans with one wave per line) ----
// A wave of a row kernel holds a PAIR of rows as one complex line, a wave of the column kernel a pair of columns;
// with row-major intermediates the column kernel touches 8 bytes of a different 128-byte line with every lane, and
// every such request moves a whole line between L2 and L1 (measured: 1.3 ms for the loads and 1.2 ms for the stores of
// shear 2 at C2, the transforms hidden behind them).  Any two lines may share a transform, so the pairs are chosen
// 64 apart -- rows (Y, Y+64), columns (X, X+64) -- and the intermediates are stored as 2x2 blocks
//     blk[t][u] = (Y1@X1, Y2@X1, Y1@X2, Y2@X2),  Y1 = row0 + 128 (t/64) + t%64,  X1 = 128 (u/64) + u%64
// (row0 = first data row r0 in A1r, = off in A2r):
// element j of a line lives in lane j%64, so both rows of a block are registers of the SAME lane of the column
// kernel and both columns registers of the same lane of the row kernels -- every access is one float4 with no
// shuffle: 16 bytes per scattered request in shear 2 (half the requests), 1 KB contiguous per instruction in shears
// 1 and 3.  A1r is indexed by DATA row (canvas row - r0; r0 = off or off + 1 after the rot90 pre-step): the kernels always
// place a frame's N live samples at the canonical line positions [off, off + N) and add the integer displacement
// (r0 - off, c0 - off) to the shift instead -- R_s x = R_{s+d} x' for x'[j] = x[j + d], exactly (the kernel is a function
// of position - shift), and sin(pi(s+d)) alt(x') = sin(pi s) alt(x).  So exactly R1/4 of the R1 inputs of every first
// butterfly are live and the pruned butterflies of fft_wave.h apply (Plan::CAN_PRUNE).
template <class P>
struct Blk {
  static constexpr int N = P::L / 4, OFF = 3 * P::L / 8;
  static constexpr int NB = N / 2;              // block rows per frame
  static constexpr int NBC = P::L / 2;          // block columns
  static constexpr int NG = N / 128;            // groups of 64 block rows
  // register of line element jb + lane (jb a multiple of 64)
  static constexpr int reg(int jb) { return ((jb % P::M1) / 64) * P::R1 + jb / P::M1; }
  static_assert(N % 128 == 0 && OFF % 64 == 0, "blocked layout needs N = L/4 a multiple of 128");
};

// affine source map of canvas'(Y, X) -> frame[base + X*stride] (rot90 folded in)
__device__ __forceinline__ void src_map(int q, int Y, const RotGeom& g, int& base, int& stride) {
  switch (q) {
    case 1: stride = g.N; base = -g.off * g.N + (g.Lc - Y - g.off); break;
    case 2: stride = -1; base = (g.Lc - Y - g.off) * g.N + (g.Lc - g.off); break;
    case 3: stride = -g.N; base = (g.Lc - g.off) * g.N + (Y - g.off); break;
    default: stride = 1; base = (Y - g.off) * g.N - g.off; break;
  }
}

// Phase factors of a shift increment delta (= 64 a for the row shears, 64 b for the column shear: the two lines of a
// pair are 64 apart): out[fl][dir][0..63] = exp(-2 pi i m_lane delta/L) (m_lane as in ShearPhase::init), [64] the factor
// of z, [65] the factor of w, [66] = (cos(pi delta), sin(pi delta)).  float64 evaluation, one tiny launch per batch.
constexpr int DPH_STRIDE = 68;
template <class P>
__global__ __launch_bounds__(128) void rs_phase_delta(const RotFrame* __restrict__ fr, int f0, cf* __restrict__ out) {
  const int fl = blockIdx.x, dir = blockIdx.y, t = threadIdx.x;
  if (t >= 67) return;
  const RotFrame p = fr[f0 + fl];
  const double delta = 64.0 * (dir == 0 ? p.a : p.b);
  double turns;
  if (t < 64) turns = (double)(t / P::R2 + P::R1 * (t % P::R2)) * delta / (double)P::L;
  else if (t == 64) turns = (double)(64 / P::R2) * delta / (double)P::L;
  else if (t == 65) turns = (double)(P::R1 * P::R2) * delta / (double)P::L;
  else turns = -0.5 * delta;                      // exp(+i pi delta) = (cos, sin)(pi delta)
  turns -= rint(turns);
  double sn, cs;
  sincospi(-2.0 * turns, &sn, &cs);
  out[((int64_t)fl * 2 + dir) * DPH_STRIDE + t] = mkcf((float)cs, (float)sn);
}

// ---- shear 1: row pairs (2p, 2p+1) ----
template <class P>
__global__ __launch_bounds__(64 * P::WPB) void rs_shear1(const float* __restrict__ in,
                                                         const RotFrame* __restrict__ fr, RotGeom g,
                                                         float* __restrict__ A1r, Aux aux, int f0, int nf,
                                                         const cf* __restrict__ twtab, int* __restrict__ counters) {
  VIPMI_SLOT_PROLOGUE();
  constexpr bool PW = false;   // one token per workgroup: its LPB line pairs are adjacent rows, which share the
                               // source frame's 64-byte sectors when the rot90 pre-step makes the gather column-wise
  constexpr bool BLK = P::WPL == 1;             // blocked intermediates, rows paired 64 apart
  // one wave per line: barrier-free workgroup sequence (RingTasks); else one token per workgroup and barrier
  using TaskSource = typename std::conditional<BLK, RingTasks<P>, Tasks<P, PW>>::type;
  TaskSource tasks;
  if constexpr (BLK) tasks.init(counters, lds_all, wave); else tasks.init(counters, lds_all);
  const int half = BLK ? Blk<P>::NB : g.N / 2;
  const int npairs = nf * half;
  constexpr int PPT = 4;                        // line pairs per token of a wave (keeps the atomics under ~25 per us)
  const int ntask = PW ? (npairs + PPT - 1) / PPT : (npairs + P::LPB - 1) / P::LPB;
  auto next_task = [&]() {
    if constexpr (BLK) {
      return tasks.next();
    } else {
      const int t = tasks.template take<1>();
      tasks.request();
      return t;
    }
  };
  if constexpr (!BLK) tasks.request();
  // One wave per line (blocked intermediates): software pipeline over the tasks of a wave -- the 16 source loads and the frame
  // record of task i + 1 are requested before the transforms of task i.  Without it the gather stands at the head of every task
  // (column-wise for half the quadrants) and, vmcnt counting loads and stores in order, the wait for it also waits for the
  // previous task's 16 KB of stores.  Measured on one box, A / B (400 frames of 512 px): 0.821 -> 0.789 ms, parity unchanged.
  if constexpr (BLK && P::CAN_PRUNE && (P::WPB == 8 || P::L == 1024)) {
    using B = Blk<P>;
    constexpr int NIN = P::U1L * P::NCNT;
    float t1n[NIN], t2n[NIN];
    RotFrame pn;
    int prn = 0;
    auto fetch = [&](int task_) {
      prn = task_ * P::LPB + slot;
      if (task_ < ntask && prn < npairs) {
        const int fl = prn / half, f = f0 + fl, tb = prn % half;
        pn = fr[f];
        const int r0 = (pn.q == 1 || pn.q == 2) ? g.alt0 : g.off;
        const int c0 = (pn.q == 2 || pn.q == 3) ? g.alt0 : g.off;
        const float* frame = in + (int64_t)f * g.N * g.N;
        const int yrel = 128 * (tb / 64) + (tb % 64);
        const int Y1 = r0 + yrel, Y2 = Y1 + 64, dc = c0 - g.off;
        int b1, st1, b2, st2;
        src_map(pn.q, Y1, g, b1, st1);
        src_map(pn.q, Y2, g, b2, st2);
#pragma unroll
        for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
          for (int n1 = 0; n1 < P::NCNT; ++n1) {
            const int X = P::M1 * (P::NLO + n1) + lane + 64 * ul + dc;
            t1n[ul * P::NCNT + n1] = frame[b1 + X * st1];
            t2n[ul * P::NCNT + n1] = frame[b2 + X * st2];
          }
      }
    };
    int task = next_task();
    fetch(task);
    while (task < ntask) {
      const int pr = prn;
      const bool live = pr < npairs;
      const RotFrame p = pn;
      cf v[P::VL];
      if (live) {
#pragma unroll
        for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
          for (int n1 = 0; n1 < P::NCNT; ++n1) {
            const float a = t1n[ul * P::NCNT + n1], b = t2n[ul * P::NCNT + n1];
            v[ul * P::R1 + P::NLO + n1] = mkcf((a == a) ? a : 0.f, (b == b) ? b : 0.f);
          }
      }
      const int nxt = next_task();
      fetch(nxt);
      if (live) {
        const int fl = pr / half, tb = pr % half;
        const int r0 = (p.q == 1 || p.q == 2) ? g.alt0 : g.off;
        const int c0 = (p.q == 2 || p.q == 3) ? g.alt0 : g.off;
        const int yrel = 128 * (tb / 64) + (tb % 64);
        const int Y1 = r0 + yrel, Y2 = Y1 + 64, dc = c0 - g.off;
        const double s1 = p.a * (double)(Y1 - g.c) + (double)dc, s2 = p.a * (double)(Y2 - g.c) + (double)dc;
        float alt1, alt2, sn1, sn2;
        pair_shift<P, true, false, true>(v, tw, lds, s1, s2, lane, sub, alt1, alt2, sn1, sn2,
                                         aux.dph + ((int64_t)fl * 2 + 0) * DPH_STRIDE);
        float4* o = reinterpret_cast<float4*>(A1r) + ((int64_t)fl * B::NB + tb) * B::NBC + lane;
#pragma unroll
        for (int gq = 0; gq < P::L / 128; ++gq) {
          const cf a = v[B::reg(128 * gq)], b = v[B::reg(128 * gq + 64)];
          o[64 * gq] = make_float4(a.x, a.y, b.x, b.y);
        }
        if (lane == 0) {
          aux.beta[fl * g.N + yrel] = sn1 * alt1;
          aux.beta[fl * g.N + yrel + 64] = sn2 * alt2;
        }
      }
      task = nxt;
    }
    return;
  }
  for (int task = next_task(); task < ntask; task = next_task()) {
   for (int sub_task = 0; sub_task < (PW ? PPT : 1); ++sub_task) {
    int pr = PW ? task * PPT + sub_task : task * P::LPB + slot;
    const bool live = pr < npairs;
    if (PW && !live) break;
    if (!live) pr = npairs - 1;
    const int fl = pr / half, f = f0 + fl;
    const RotFrame p = fr[f];
    const int r0 = (p.q == 1 || p.q == 2) ? g.alt0 : g.off;
    const int c0 = (p.q == 2 || p.q == 3) ? g.alt0 : g.off;
    const int tb = pr % half;                                       // block row (BLK)
    const float* frame = in + (int64_t)f * g.N * g.N;
    cf v[P::VL];
    if constexpr (BLK) {
      // data rows yrel, yrel + 64 (canvas rows r0 + yrel); the N live columns [c0, c0 + N) are placed at the canonical
      // positions [off, off + N) and the displacement c0 - off goes into the shift (see Blk)
      constexpr bool PR = P::CAN_PRUNE;
      if (!live) continue;                                          // (one wave per line: no barrier inside)
      const int yrel = 128 * (tb / 64) + (tb % 64);
      const int Y1 = r0 + yrel, Y2 = Y1 + 64, dc = c0 - g.off;
      int b1, st1, b2, st2;
      src_map(p.q, Y1, g, b1, st1);
      src_map(p.q, Y2, g, b2, st2);
      if constexpr (!PR) {
#pragma unroll
        for (int i = 0; i < P::VL; ++i) v[i] = mkcf(0.f, 0.f);
      }
#pragma unroll
      for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
        for (int n1 = P::NLO; n1 < P::NLO + P::NCNT; ++n1) {
          const int X = P::M1 * n1 + lane + 64 * ul + dc;           // canvas column of canonical position M1 n1 + ...
          const float t1 = frame[b1 + X * st1], t2 = frame[b2 + X * st2];
          v[ul * P::R1 + n1] = mkcf((t1 == t1) ? t1 : 0.f, (t2 == t2) ? t2 : 0.f);
        }
      const double s1 = p.a * (double)(Y1 - g.c) + (double)dc, s2 = p.a * (double)(Y2 - g.c) + (double)dc;
      float alt1, alt2, sn1, sn2;
      pair_shift<P, PR, false, true>(v, tw, lds, s1, s2, lane, sub, alt1, alt2, sn1, sn2,
                               aux.dph + ((int64_t)fl * 2 + 0) * DPH_STRIDE);
      using B = Blk<P>;
      float4* o = reinterpret_cast<float4*>(A1r) + ((int64_t)fl * B::NB + tb) * B::NBC + lane;
#pragma unroll
      for (int gq = 0; gq < P::L / 128; ++gq) {
        const cf a = v[B::reg(128 * gq)], b = v[B::reg(128 * gq + 64)];
        o[64 * gq] = make_float4(a.x, a.y, b.x, b.y);
      }
      if (lane == 0) {                          // sin(pi(s+dc)) alt(x') = sin(pi s) alt(x): no sign to restore
        aux.beta[fl * g.N + yrel] = sn1 * alt1;
        aux.beta[fl * g.N + yrel + 64] = sn2 * alt2;
      }
    } else {
      constexpr bool PR = P::CAN_PRUNE;
      const int Y1 = r0 + 2 * tb, Y2 = Y1 + 1, dc = c0 - g.off;
      const int yrel = Y1 - r0;
      int b1, st1, b2, st2;
      src_map(p.q, Y1, g, b1, st1);
      src_map(p.q, Y2, g, b2, st2);
      if constexpr (!PR) {
#pragma unroll
        for (int i = 0; i < P::VL; ++i) v[i] = mkcf(0.f, 0.f);
      }
#pragma unroll
      for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
        for (int n1 = P::NLO; n1 < P::NLO + P::NCNT; ++n1) {      // the N live columns at the canonical positions (see Blk)
          const int X = P::M1 * n1 + lane + 64 * (sub * P::U1L + ul) + dc;
          const float t1 = frame[b1 + X * st1], t2 = frame[b2 + X * st2];
          v[ul * P::R1 + n1] = mkcf((t1 == t1) ? t1 : 0.f, (t2 == t2) ? t2 : 0.f);
        }
      const double s1 = p.a * (double)(Y1 - g.c) + (double)dc, s2 = p.a * (double)(Y2 - g.c) + (double)dc;
      float alt1, alt2, sn1, sn2;
      pair_shift<P, PR, false>(v, tw, lds, s1, s2, lane, sub, alt1, alt2, sn1, sn2);
      if (live) {
        float* o1 = A1r + ((int64_t)fl * g.N + yrel) * P::L;
        float* o2 = o1 + P::L;
#pragma unroll
        for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
          for (int n1 = 0; n1 < P::R1; ++n1) {
            const int X = P::M1 * n1 + lane + 64 * (sub * P::U1L + ul);
            o1[X] = v[ul * P::R1 + n1].x;
            o2[X] = v[ul * P::R1 + n1].y;
          }
        if (sub == 0 && lane == 0) {
          aux.beta[fl * g.N + yrel] = sn1 * alt1;
          aux.beta[fl * g.N + yrel + 1] = sn2 * alt2;
        }
      }
    }
   }
  }
}

// Bf[f] = sum_Y (-1)^Y beta_Y   (Y = canvas row of data row yrel)
__global__ __launch_bounds__(256) void rs_bf_kernel(const RotFrame* __restrict__ fr, RotGeom g, Aux aux, int f0) {
  __shared__ float sh[4];
  const int fl = blockIdx.x;
  const RotFrame p = fr[f0 + fl];
  const int r0 = (p.q == 1 || p.q == 2) ? g.alt0 : g.off;
  float s = 0.f;
  for (int y = threadIdx.x; y < g.N; y += blockDim.x) {
    const float b = aux.beta[fl * g.N + y];
    s += ((r0 + y) & 1) ? -b : b;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) aux.bf[fl] = sh[0] + sh[1] + sh[2] + sh[3];
}

// K[fl][m] = ifft( fft(beta~) g )[off + m]: one line per frame
template <class P>
__global__ __launch_bounds__(64 * P::WPB) void rs_aux_k(const RotFrame* __restrict__ fr, RotGeom g, Aux aux, int f0,
                                                        int nf, const cf* __restrict__ twtab) {
  VIPMI_SLOT_PROLOGUE();
  constexpr int R1 = P::R1, R2 = P::R2, R3 = P::R3;
  const int niter = (nf + gridDim.x * P::LPB - 1) / (gridDim.x * P::LPB);
  for (int it = 0; it < niter; ++it) {
    int fl = (it * gridDim.x + blockIdx.x) * P::LPB + slot;
    const bool live = fl < nf;
    if (!live) fl = nf - 1;
    const RotFrame p = fr[f0 + fl];
    const int r0 = (p.q == 1 || p.q == 2) ? g.alt0 : g.off;
    cf v[P::VL];
#pragma unroll
    for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
      for (int n1 = 0; n1 < P::R1; ++n1) {
        float x = 0.f;
        if (n1 >= P::NLO && n1 <= P::NLO + P::NCNT) {
          const int yrel = P::M1 * n1 + lane + 64 * (sub * P::U1L + ul) - r0;
          if (yrel >= 0 && yrel < g.N) x = aux.beta[fl * g.N + yrel];
        }
        v[ul * P::R1 + n1] = mkcf(x, 0.f);
      }
    // the zero imaginary parts stay opaque: folded into the first butterflies they made hipcc emit packed adds of a register pair
    // with its own swapped self in the src1-high operand form (common.h, VIPMI_NO_PK32; tools/isa_lint.py)
#pragma unroll
    for (int i = 0; i < P::VL; ++i) asm volatile("" : "+v"(v[i]));
    fft_forward<P>(v, tw, lds, lane, sub);
    // g(k) = sum_{X=0}^{L-1} exp(-2 pi i ks b (X - c)/L) = exp(-2 pi i ks b ((L-1)/2 - c)/L) sin(pi ks b)/sin(pi ks b/L)
    const int u0 = sub * P::U3L;
#pragma unroll
    for (int ul = 0; ul < P::U3L; ++ul) {
      const int wq = lane + 64 * (u0 + ul), k1 = wq / R2, ka = wq % R2;
#pragma unroll
      for (int kb = 0; kb < R3; ++kb) {
        const int kbs = (kb < R3 / 2) ? kb : kb - R3;
        const double kbv = (double)(k1 + R1 * ka + R1 * R2 * kbs) * p.b;
        const double den = sinpi(kbv / (double)P::L);
        const double ratio = (fabs(den) < 1e-300) ? (double)P::L : sinpi(kbv) / den;
        double sn, cs;
        sincospi(-2.0 * kbv * (0.5 * (double)(P::L - 1) - (double)g.c) / (double)P::L, &sn, &cs);
        float gre = (float)(ratio * cs / (double)P::L), gim = (float)(ratio * sn / (double)P::L);
        if (kb == R3 / 2 && wq == 0) gim = 0.f;          // Nyquist: sum of cos(pi s_X) (real part)
        v[ul * R3 + kb] = cmul(v[ul * R3 + kb], mkcf(gre, gim));   // (the hand-written form: hipcc's own packed lowering of
                                                                   // the scalar formula used the src1-high operand form, common.h)
      }
    }
    fft_inverse<P>(v, tw, lds, lane, sub);
    if (live) {
#pragma unroll
      for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
        for (int n1 = P::NLO; n1 < P::NLO + P::NCNT; ++n1) {
          const int m = P::M1 * (n1 - P::NLO) + lane + 64 * (sub * P::U1L + ul);
          aux.kv[fl * g.N + m] = v[ul * P::R1 + n1].x;
        }
    }
  }
}

// ---- shear 2: column pairs; a workgroup handles 2*LPB adjacent columns, tile staged through LDS ----
template <class P>
__global__ __launch_bounds__(64 * P::WPB) void rs_shear2(const float* __restrict__ A1r,
                                                         const RotFrame* __restrict__ fr, RotGeom g,
                                                         float* __restrict__ A2r, Aux aux, int f0, int nf,
                                                         const cf* __restrict__ twtab, int* __restrict__ counters) {
  VIPMI_SLOT_PROLOGUE();
  constexpr int W = 2 * P::LPB, LDT = W + 1;      // tile row stride in floats (odd: conflict-free column reads)
  float* tile = reinterpret_cast<float*>(lds_all);   // [N][LDT], aliases the exchange regions between phases
  const int groups = (P::L + W - 1) / W;          // last group may be ragged (W need not divide L)
  const int units = nf * groups;
  // Two adjacent column groups share every 128-byte line of A1r / A2r (a group is 64 or 48 bytes wide): consecutive
  // tokens of an XCD are the two halves of one such pair, so they are taken by two workgroups of the same XCD at
  // about the same time and the second half of each line is an L2 hit instead of a second HBM fetch.
  Tasks<P, false> tasks;
  tasks.init(counters, lds_all);
  tasks.request();
  for (int unit = tasks.template take<2>(); unit < units; unit = tasks.template take<2>()) {
    {
      tasks.request();
      const int fl = unit / groups, X0 = (unit % groups) * W, f = f0 + fl;
      const RotFrame p = fr[f];
      const int r0 = (p.q == 1 || p.q == 2) ? g.alt0 : g.off;
      const float* src = A1r + (int64_t)fl * g.N * P::L + X0;
      const int wcols = (P::L - X0 < W) ? (P::L - X0) : W;
      for (int e = threadIdx.x; e < g.N * W; e += 64 * P::WPB) {
        const int row = e / W, c = e % W;
        tile[row * LDT + c] = (c < wcols) ? src[(int64_t)row * P::L + c] : 0.f;
      }
      __syncthreads();
      constexpr bool PR = P::CAN_PRUNE;
      const int dr = r0 - g.off;                   // data rows at the canonical positions, displacement -> shift (see Blk)
      cf v[P::VL];
      if constexpr (!PR) {
#pragma unroll
        for (int i = 0; i < P::VL; ++i) v[i] = mkcf(0.f, 0.f);
      }
#pragma unroll
      for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
        for (int n1 = P::NLO; n1 < P::NLO + P::NCNT; ++n1) {
          const int yrel = P::M1 * (n1 - P::NLO) + lane + 64 * (sub * P::U1L + ul);      // off == M1*NLO
          v[ul * P::R1 + n1] = mkcf(tile[yrel * LDT + 2 * slot], tile[yrel * LDT + 2 * slot + 1]);
        }
      __syncthreads();
      const int X1 = X0 + 2 * slot, X2 = X1 + 1;
      const double s1 = p.b * (double)(X1 - g.c) + (double)dr, s2 = p.b * (double)(X2 - g.c) + (double)dr;
      float alt1, alt2, sn1, sn2;
      pair_shift<P, PR, PR>(v, tw, lds, s1, s2, lane, sub, alt1, alt2, sn1, sn2);
      __syncthreads();
      // rank-one correction  - sin(pi s_X) (-1)^X Bf/L (-1)^Y  on the output rows Y = off + m;  sn = sin(pi(s+dr))/L
      const float bfl = (dr & 1) ? -aux.bf[fl] : aux.bf[fl];
      const float k1c = sn1 * ((X1 & 1) ? -bfl : bfl);
      const float k2c = sn2 * ((X2 & 1) ? -bfl : bfl);
#pragma unroll
      for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
        for (int n1 = P::NLO; n1 < P::NLO + P::NCNT; ++n1) {
          const int m = P::M1 * (n1 - P::NLO) + lane + 64 * (sub * P::U1L + ul);      // off == M1*NLO
          const float sg = ((g.off + m) & 1) ? -1.f : 1.f;
          tile[m * LDT + 2 * slot] = v[ul * P::R1 + n1].x - sg * k1c;
          tile[m * LDT + 2 * slot + 1] = v[ul * P::R1 + n1].y - sg * k2c;
        }
      if (sub == 0 && lane == 0 && X1 < P::L) {
        aux.gam[fl * P::L + X1] = ((X1 & 1) ? -sn1 : sn1) * alt1;
        aux.gam[fl * P::L + X2] = ((X2 & 1) ? -sn2 : sn2) * alt2;
      }
      __syncthreads();
      float* dst = A2r + (int64_t)fl * g.N * P::L + X0;
      for (int e = threadIdx.x; e < g.N * W; e += 64 * P::WPB) {
        const int row = e / W, c = e % W;
        if (c < wcols) dst[(int64_t)row * P::L + c] = tile[row * LDT + c];
      }
      __syncthreads();
    }
  }
}

// ---- shear 2 without the LDS tile (one wave per line, blocked intermediates -- see Blk): every wave reads and writes
// its two columns (X, X + 64) directly, one 2x2 block = 16 bytes per lane.  A load instruction then touches 64 different
// 128-byte lines, but the 8 block columns that share those lines are consecutive tokens of one XCD queue, so they are
// processed by waves of the same XCD at about the same time: one HBM fetch per line, the rest L2 hits; the partial
// stores merge in the L2 the same way.  No workgroup barrier and no global <-> LDS staging phase in which all eight
// waves of a workgroup wait (36 % of the tiled kernel).  Measured at C2 (400 x 512^2, 2.6 ms): transforms alone
// 2.2 ms, loads alone 0.9 ms, stores alone 0.9 ms (with 8-byte row-major accesses: 1.3 + 1.2 ms) -- the kernel is bound
// by the transforms' instruction stream (2443 VALU instructions per line pair, 69 % of them packed FP32 arithmetic,
// issued at 75 % of the SIMD rate with two waves per SIMD), not by memory.
template <class P>
__global__ __launch_bounds__(64 * P::WPB) void rs_shear2_direct(const float* __restrict__ A1r,
                                                                const RotFrame* __restrict__ fr, RotGeom g,
                                                                float* __restrict__ A2r, Aux aux, int f0, int nf,
                                                                const cf* __restrict__ twtab,
                                                                int* __restrict__ counters) {
  static_assert(P::WPL == 1, "rs_shear2_direct: one wave per line");
  using B = Blk<P>;
  VIPMI_SLOT_PROLOGUE();
  Tasks<P, true> tasks;
  tasks.init(counters, lds_all);
  constexpr int TPG = 8;                          // 8 block columns = one 128-byte line of every block row
  const int ntask = nf * B::NBC;
  constexpr bool PR = P::CAN_PRUNE;
  // Software pipeline over the tasks of a wave (the plan with ONE wave per SIMD -- nothing else covers the load latency
  // at the head of a task there -- and Le = 1024, -2.6 %; at Le = 2048 it changed nothing, NOTES.md): the input blocks, the frame
  // record and Bf of task i+1 are requested before the transforms of task i.  Tokens then run two ahead.
  constexpr bool PIPE = P::WPB <= 4 || P::L == 1024;
  float4 nblk[B::NG];
  RotFrame pn;
  float bfn = 0.f;
  auto fetch = [&](int t) {
    if (t < ntask) {
      const int fl = t / B::NBC, u = t % B::NBC;
      const float4* src = reinterpret_cast<const float4*>(A1r) + (int64_t)fl * B::NB * B::NBC + u;
#pragma unroll
      for (int G = 0; G < B::NG; ++G) nblk[G] = src[(int64_t)(64 * G + lane) * B::NBC];
      pn = fr[f0 + fl];
      bfn = aux.bf[fl];
    }
  };
  tasks.request();
  int task = tasks.template take<TPG>();
  tasks.request();
  if constexpr (PIPE) fetch(task);
  while (task < ntask) {
    if constexpr (!PIPE) fetch(task);
    const int fl = task / B::NBC, u = task % B::NBC;
    const int X1 = 128 * (u / 64) + (u % 64), X2 = X1 + 64;
    const RotFrame p = pn;
    const float bf0 = bfn;
    const int r0 = (p.q == 1 || p.q == 2) ? g.alt0 : g.off;
    const int dr = r0 - g.off;                     // the data rows sit at the canonical positions: displacement -> shift
    cf v[P::VL];
    if constexpr (!PR) {
#pragma unroll
      for (int i = 0; i < P::VL; ++i) v[i] = mkcf(0.f, 0.f);
    }
#pragma unroll
    for (int G = 0; G < B::NG; ++G) {
      const float4 blk = nblk[G];
      v[B::reg(B::OFF + 128 * G)] = mkcf(blk.x, blk.z);
      v[B::reg(B::OFF + 128 * G + 64)] = mkcf(blk.y, blk.w);
    }
    int next = ntask;
    if constexpr (PIPE) {
      next = tasks.template take<TPG>();
      tasks.request();
      fetch(next);
    }
    const double s1 = p.b * (double)(X1 - g.c) + (double)dr, s2 = p.b * (double)(X2 - g.c) + (double)dr;
    float alt1, alt2, sn1, sn2;
    pair_shift<P, PR, PR, true>(v, tw, lds, s1, s2, lane, sub, alt1, alt2, sn1, sn2,
                          aux.dph + ((int64_t)fl * 2 + 1) * DPH_STRIDE);
    // rank-one correction  - sin(pi s_X) (-1)^X Bf/L (-1)^Y  on the output rows Y = off + m;  sn = sin(pi(s+dr))/L
    const float bfl = (dr & 1) ? -bf0 : bf0;
    const float k1c = sn1 * ((X1 & 1) ? -bfl : bfl);
    const float k2c = sn2 * ((X2 & 1) ? -bfl : bfl);
    // A lane's 16-byte block shares its 32-byte sector with the block of the NEXT column pair, which another wave writes
    // some time later: two thirds of the sectors go to HBM twice (2.6 GB written for a 1.68 GB intermediate).  Paired layout
    // (option rot_pair_store = 1, even NG): the blocks of row groups G and G + 1 of one column pair lie side by side -- the
    // lane writes 32 contiguous bytes back to back, shear 3 reads them in two consecutive sub-tasks of one wave.  Measured
    // (C2, round 3): WRITE_SIZE of this kernel 2.60 -> 2.02 GB, but FETCH_SIZE of shear 3 1.04 -> 1.85 GB (the second
    // sub-task finds its half of the lines evicted: 16 MB per XCD are in flight between the two), times unchanged (3.82 /
    // 3.81 ms; 1024 px: 13.5 / 13.8 ms): neither kernel is bound by memory.  Off by default.
    float4* dst = reinterpret_cast<float4*>(A2r) + (int64_t)fl * B::NB * B::NBC;
#pragma unroll
    for (int G = 0; G < B::NG; ++G) {
      const cf a = v[B::reg(B::OFF + 128 * G)], b = v[B::reg(B::OFF + 128 * G + 64)];
      const float sg = ((B::OFF + lane) & 1) ? -1.f : 1.f;           // rows Y and Y + 64 (+128 G): same parity
      const int64_t at = aux.pair ? ((int64_t)(64 * (G >> 1) + lane) * B::NBC + u) * 2 + (G & 1)
                                  : (int64_t)(64 * G + lane) * B::NBC + u;
      dst[at] = make_float4(a.x - sg * k1c, b.x - sg * k1c, a.y - sg * k2c, b.y - sg * k2c);
    }
    if (lane == 0) {
      aux.gam[fl * P::L + X1] = ((X1 & 1) ? -sn1 : sn1) * alt1;
      aux.gam[fl * P::L + X2] = ((X2 & 1) ? -sn2 : sn2) * alt2;
    }
    if constexpr (PIPE) {
      task = next;
    } else {
      task = tasks.template take<TPG>();
      tasks.request();
    }
  }
}

// Gam[f] = sum_X (-1)^X gamma_X  (fixed-order tree: deterministic)
__global__ __launch_bounds__(256) void rs_gamma_kernel(Aux aux, int L) {
  __shared__ float sh[4];
  const int fl = blockIdx.x;
  float s = 0.f;
  for (int x = threadIdx.x; x < L; x += blockDim.x) s += aux.gam[fl * L + x];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) aux.gsum[fl] = sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- shear 3: row pairs of A2r -> final real frame rows, crop, mask restore ----
template <class P>
__global__ __launch_bounds__(64 * P::WPB) void rs_shear3(const float* __restrict__ A2r,
                                                         const RotFrame* __restrict__ fr, RotGeom g,
                                                         const float* __restrict__ in, float* __restrict__ out,
                                                         Aux aux, int f0, int nf, int mask_nan, int mask_zero, float mask_v,
                                                         const cf* __restrict__ twtab, int* __restrict__ counters) {
  VIPMI_SLOT_PROLOGUE();
  constexpr bool PW = P::WPL == 1;
  Tasks<P, PW> tasks;
  tasks.init(counters, lds_all);
  constexpr bool BLK = P::WPL == 1;             // blocked intermediates, rows paired 64 apart
  const int half = g.N / 2;
  const int npairs = nf * half;
  constexpr int PPT = 4;                        // line pairs per token of a wave (keeps the atomics under ~25 per us)
  const int ntask = PW ? (npairs + PPT - 1) / PPT : (npairs + P::LPB - 1) / P::LPB;
  tasks.request();
  for (int task = tasks.template take<1>(); task < ntask; task = tasks.template take<1>()) {
    tasks.request();
   for (int sub_task = 0; sub_task < (PW ? PPT : 1); ++sub_task) {
    int pr = PW ? task * PPT + sub_task : task * P::LPB + slot;
    const bool live = pr < npairs;
    if (PW && !live) break;
    if (!live) pr = npairs - 1;
    const int fl = pr / half, f = f0 + fl;
    int tb = pr % half;
    if (BLK && aux.pair) tb = 64 * (2 * (tb >> 7) + (tb & 1)) + ((tb >> 1) & 63);     // (consecutive pairs: the two halves of a 32-byte block pair)
    const int m = BLK ? 128 * (tb / 64) + (tb % 64) : 2 * tb, dm = BLK ? 64 : 1;     // output rows m, m + dm
    const RotFrame p = fr[f];
    const int Y1 = g.off + m, Y2 = Y1 + dm;
    cf v[P::VL];
    if constexpr (BLK) {
      using B = Blk<P>;
      const int Gt = tb >> 6, lt = tb & 63;
      const float4* ib = aux.pair ? reinterpret_cast<const float4*>(A2r) + (((int64_t)fl * (B::NB / 2) + 64 * (Gt >> 1) + lt) * B::NBC + lane) * 2 + (Gt & 1)
                                  : reinterpret_cast<const float4*>(A2r) + ((int64_t)fl * B::NB + tb) * B::NBC + lane;
      const int gstep = aux.pair ? 128 : 64;
#pragma unroll
      for (int gq = 0; gq < P::L / 128; ++gq) {
        const float4 blk = ib[gstep * gq];
        v[B::reg(128 * gq)] = mkcf(blk.x, blk.y);
        v[B::reg(128 * gq + 64)] = mkcf(blk.z, blk.w);
      }
    } else {
      const float* i1 = A2r + ((int64_t)fl * g.N + m) * P::L;
      const float* i2 = i1 + P::L;
#pragma unroll
      for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
        for (int n1 = 0; n1 < P::R1; ++n1) {
          const int X = P::M1 * n1 + lane + 64 * (sub * P::U1L + ul);
          v[ul * P::R1 + n1] = mkcf(i1[X], i2[X]);
        }
    }
    const double s1 = p.a * (double)(Y1 - g.c), s2 = p.a * (double)(Y2 - g.c);
    // the source pixels that restore the mask at the end are requested NOW (16 registers): asked for after the transforms, their
    // round trip stood at the tail of every task
    const int64_t ob1 = ((int64_t)f * g.N + m) * g.N, ob2 = ob1 + (int64_t)dm * g.N;
    const float gs = aux.gsum[fl], kv1 = aux.kv[fl * g.N + m], kv2 = aux.kv[fl * g.N + m + dm];     // (likewise)
    constexpr bool EARLY = BLK && P::L <= 2048;            // (the Le = 4096 plans have no registers to spare)
    float srcA[P::U1L * P::NCNT], srcB[P::U1L * P::NCNT];
    if constexpr (EARLY) {
#pragma unroll
      for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
        for (int n1 = 0; n1 < P::NCNT; ++n1) {
          const int j = P::M1 * n1 + lane + 64 * (sub * P::U1L + ul);
          srcA[ul * P::NCNT + n1] = in[ob1 + j];
          srcB[ul * P::NCNT + n1] = in[ob2 + j];
        }
    }
    float alt1, alt2, sn1, sn2;
    pair_shift<P, false, P::CAN_PRUNE, BLK>(v, tw, lds, s1, s2, lane, sub, alt1, alt2, sn1, sn2,
                                                   aux.dph + ((int64_t)fl * 2 + 0) * DPH_STRIDE);
    if (live) {
      const float c1 = sn1 * (kv1 + ((Y1 & 1) ? -gs : gs));
      const float c2 = sn2 * (kv2 + ((Y2 & 1) ? -gs : gs));
#pragma unroll
      for (int ul = 0; ul < P::U1L; ++ul)
#pragma unroll
        for (int n1 = P::NLO; n1 < P::NLO + P::NCNT; ++n1) {
          const int j = P::M1 * (n1 - P::NLO) + lane + 64 * (sub * P::U1L + ul);        // off == M1*NLO
          const float sg = ((g.off + j) & 1) ? -1.f : 1.f;
          float re1 = v[ul * P::R1 + n1].x - sg * c1;
          float re2 = v[ul * P::R1 + n1].y - sg * c2;
          const float src1 = EARLY ? srcA[ul * P::NCNT + (n1 - P::NLO)] : in[ob1 + j];
          const float src2 = EARLY ? srcB[ul * P::NCNT + (n1 - P::NLO)] : in[ob2 + j];
          if (mask_nan && !(src1 == src1)) re1 = __uint_as_float(0x7fc00000u);
          if (mask_nan && !(src2 == src2)) re2 = __uint_as_float(0x7fc00000u);
          if (mask_zero && src1 == mask_v) re1 = mask_v;
          if (mask_zero && src2 == mask_v) re2 = mask_v;
          out[ob1 + j] = re1;
          out[ob2 + j] = re2;
        }
    }
   }
  }
}

template <class P>
int run_plan2(vipmi_ctx* ctx, const float* in, const RotFrame* d_frames, const RotGeom& g, int64_t n, float* out,
              int mask_nan, int mask_zero, float mask_v) {
  constexpr bool BLK = P::WPL == 1;                         // blocked intermediates (see Blk)
  VIPMI_REQUIRE(4 * g.N == P::L && 8 * g.off == 3 * P::L, "derotate(fft2): unexpected canvas geometry");
  const int64_t per_frame = (int64_t)(BLK ? g.N + 2 : g.N) * P::L;     // floats per intermediate per frame
  int64_t chunk = ctx->opt("rot_batch", 0);
  if (chunk <= 0) {
    int64_t budget = ctx->opt("rot_ws_mb", 4096) * (int64_t)(1 << 20);
    chunk = budget / (2 * per_frame * (int64_t)sizeof(float));
  }
  if (chunk < 1) chunk = 1;
  if (chunk > n) chunk = n;
  float *A1r = nullptr, *A2r = nullptr;
  VIPMI_TRY(ws(ctx, "rot_a1", (size_t)(chunk * per_frame), &A1r));
  VIPMI_TRY(ws(ctx, "rot_a2", (size_t)(chunk * per_frame), &A2r));
  Aux aux;
  VIPMI_TRY(ws(ctx, "rot_beta", (size_t)(chunk * g.N), &aux.beta));
  VIPMI_TRY(ws(ctx, "rot_bf", (size_t)chunk, &aux.bf));
  VIPMI_TRY(ws(ctx, "rot_kv", (size_t)(chunk * g.N), &aux.kv));
  VIPMI_TRY(ws(ctx, "rot_gam", (size_t)(chunk * P::L), &aux.gam));
  VIPMI_TRY(ws(ctx, "rot_gsum", (size_t)chunk, &aux.gsum));
  VIPMI_TRY(ws(ctx, "rot_dph", (size_t)chunk * 2 * DPH_STRIDE, &aux.dph));
  aux.pair = (BLK && (Blk<P>::NG % 2 == 0) && ctx->opt("rot_pair_store", 0) != 0) ? 1 : 0;
  int* counters = nullptr;                       // 3 kernels x 8 task queues, one 128-byte line each
  VIPMI_TRY(ws(ctx, "rot_counters", (size_t)3 * 256, &counters));
  size_t lds = (size_t)P::LPB * P::LDS_ELEMS * sizeof(cf);
  const size_t tile = (size_t)g.N * (2 * P::LPB + 1) * sizeof(float);
  VIPMI_REQUIRE(tile <= lds, "derotate(fft2): staging tile larger than the exchange regions");
  lds += (size_t)Twiddles<P>::LDS_ELEMS * sizeof(cf);
  VIPMI_REQUIRE(lds <= 160 * 1024, "derotate(fft2): LDS budget exceeded (%zu)", lds);
  cf* twtab = nullptr;
  {
    char key[32];
    snprintf(key, sizeof key, "L%d", P::L);
    void* pt = nullptr;
    if (!ctx->cached("rot_twiddles", key, &pt)) {
      std::vector<cf> tab;
      Twiddles<P>::fill_table(tab);
      VIPMI_TRY(ctx->upload_cached("rot_twiddles", key, tab.data(), tab.size() * sizeof(cf), &pt));
    }
    twtab = reinterpret_cast<cf*>(pt);
  }
  // (the one-line-per-frame K kernel of the one-wave Le = 4096 plan would spill: it runs on the two-wave plan of the same
  // radices -- same twiddle table, same LDS footprint, same number of lines per workgroup)
  // (Le = 2048: with eight lines per workgroup the float64 phase factors of the kernel spilled 105 registers under the 256
  // budget; it runs on the four-line plan of the same radices -- one wave per SIMD, 512 registers: 49 -> 24 us)
  using PA = typename std::conditional<std::is_same<P, Plan4096w1>::value, Plan4096w2,
                                       typename std::conditional<std::is_same<P, Plan2048w1>::value, Plan2048w1h, P>::type>::type;
  static_assert(PA::LDS_ELEMS == P::LDS_ELEMS && Twiddles<PA>::LDS_ELEMS == Twiddles<P>::LDS_ELEMS && PA::LPB <= P::LPB,
                "rs_aux_k: substitute plan must fit the launch geometry");
  auto k1 = rs_shear1<P>;
  auto ka = rs_aux_k<PA>;
  auto k2 = rs_shear2<P>;
  auto k3 = rs_shear3<P>;
  for (const void* f : {reinterpret_cast<const void*>(k1), reinterpret_cast<const void*>(ka),
                        reinterpret_cast<const void*>(k2), reinterpret_cast<const void*>(k3)})
    VIPMI_CHECK_HIP(set_dyn_lds(f, (int)lds));
  const int wgs_per_cu = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;
  int usable_cu = ctx->num_cu - (int)ctx->opt("reserve_cus", 0);
  if (usable_cu < 1) usable_cu = 1;
  const int maxwg = usable_cu * wgs_per_cu;
  const dim3 blk(64 * P::WPB);
  for (int64_t f0 = 0; f0 < n; f0 += chunk) {
    const int nf = (int)((n - f0) < chunk ? (n - f0) : chunk);
    const int64_t npairs = (int64_t)nf * (g.N / 2);
    int gr = (int)cdiv(npairs, P::LPB);
    if (gr > maxwg) gr = maxwg;
    if (gr < 8) gr = 8;
    gr = gr / 8 * 8;
    const int64_t units = (int64_t)nf * ((P::L + 2 * P::LPB - 1) / (2 * P::LPB));
    int gc = (int)(units < maxwg ? units : maxwg);
    // tasks are dealt through 8 queues (queue = workgroup index mod 8): every queue needs a workgroup, and equal
    // numbers of them
    if (gc < 8) gc = 8;
    gc = gc / 8 * 8;
    int ga = (int)cdiv(nf, PA::LPB);
    if (ga > maxwg) ga = maxwg;
    VIPMI_CHECK_HIP(hipMemsetAsync(counters, 0, 3 * 256 * sizeof(int), ctx->stream));
    if constexpr (BLK)
      hipLaunchKernelGGL(rs_phase_delta<P>, dim3(nf, 2), dim3(128), 0, ctx->stream, d_frames, (int)f0, aux.dph);
    ctx->tic("k_rot_s1");
    hipLaunchKernelGGL(k1, dim3(gr), blk, lds, ctx->stream, in, d_frames, g, A1r, aux, (int)f0, nf, twtab, counters);
    ctx->toc("k_rot_s1");
    ctx->tic("k_rot_aux");
    hipLaunchKernelGGL(rs_bf_kernel, dim3(nf), dim3(256), 0, ctx->stream, d_frames, g, aux, (int)f0);
    hipLaunchKernelGGL(ka, dim3(ga), dim3(64 * PA::WPB), lds, ctx->stream, d_frames, g, aux, (int)f0, nf, twtab);
    ctx->toc("k_rot_aux");
    ctx->tic("k_rot_s2");
    if constexpr (P::WPL == 1) {
      auto k2d = rs_shear2_direct<P>;
      VIPMI_CHECK_HIP(set_dyn_lds(reinterpret_cast<const void*>(k2d), (int)lds));
      hipLaunchKernelGGL(k2d, dim3(gr), blk, lds, ctx->stream, A1r, d_frames, g, A2r, aux, (int)f0, nf, twtab,
                         counters + 256);
    } else {
      hipLaunchKernelGGL(k2, dim3(gc), blk, lds, ctx->stream, A1r, d_frames, g, A2r, aux, (int)f0, nf, twtab,
                         counters + 256);
    }
    ctx->toc("k_rot_s2");
    ctx->tic("k_rot_aux");
    hipLaunchKernelGGL(rs_gamma_kernel, dim3(nf), dim3(256), 0, ctx->stream, aux, P::L);
    ctx->toc("k_rot_aux");
    ctx->tic("k_rot_s3");
    hipLaunchKernelGGL(k3, dim3(gr), blk, lds, ctx->stream, A2r, d_frames, g, in, out, aux, (int)f0, nf, mask_nan,
                       mask_zero, mask_v, twtab, counters + 512);
    ctx->toc("k_rot_s3");
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  return VIPMI_OK;
}

}  // namespace

int derotate_fft2(vipmi_ctx* ctx, const float* in, const RotFrame* d_frames, const RotGeom& g, int64_t n,
                  float* out, int mask_nan, int mask_zero, float mask_v) {
  switch (g.Le) {
    case 512: return run_plan2<Plan512>(ctx, in, d_frames, g, n, out, mask_nan, mask_zero, mask_v);
    // one plan per padded length.  Measured and dropped from the build (NOTES.md): 4-wave workgroups at Le = 1024 (+4 %
    // stand-alone, but with two calls in flight the other call's eigensolver finds no free CU: C4 +11 %); two waves per
    // line at Le = 2048 with 12 or 16 waves per workgroup (3 waves per SIMD, but workgroup barriers and 1.5x the
    // instructions per line: 6.2 against 3.8 ms at C2); four waves per line at Le = 4096.
    case 1024:
      // four lines per workgroup, three workgroups per CU = THREE waves per SIMD (the Le = 1024 kernels hold <= 164 VGPRs; with
      // eight-wave workgroups only one fits a CU).  Measured (round 5, A / B on one box): 1600 frames of 256 px shear 2 2.46 -> 2.35 and
      // 2.52 -> 2.31 ms, shear 1 -5 %, shear 3 -2 %; C4 24.8 -> 24.2 ms.  The chip sits at its 1400 W cap either way,
      // which is why a third wave buys 5 % and not the 15 % its idle issue slots suggest.  rot_1024_q = 0: the eight-wave plan.
      if (ctx->opt("rot_1024_q", 1)) return run_plan2<Plan1024q>(ctx, in, d_frames, g, n, out, mask_nan, mask_zero, mask_v);
      return run_plan2<Plan1024>(ctx, in, d_frames, g, n, out, mask_nan, mask_zero, mask_v);
    case 2048:
      // (four-wave workgroups, one wave per SIMD, so that the MFMA-bound Gram of another call could share the SIMDs: measured in
      // round 3, 3.90 -> 4.55 ms alone and 5.60 -> 6.20 ms pipelined, removed from the build -- DESIGN 7.1)
      return run_plan2<Plan2048w1>(ctx, in, d_frames, g, n, out, mask_nan, mask_zero, mask_v);
    case 4096:
      // one wave per line and per SIMD (512-VGPR budget), column shear software-pipelined: 200 frames of 1024 px
      // 10.85 -> 9.4 ms against the two-wave plan (kept behind rot_4096_w1=0)
      if (ctx->opt("rot_4096_w1", 1)) return run_plan2<Plan4096w1>(ctx, in, d_frames, g, n, out, mask_nan, mask_zero, mask_v);
      return run_plan2<Plan4096w2>(ctx, in, d_frames, g, n, out, mask_nan, mask_zero, mask_v);
    default:
      set_error("derotate(fft2): unsupported padded length %d", g.Le);
      return VIPMI_ERR_UNSUPPORTED;
  }
}

}  // namespace vipmi
