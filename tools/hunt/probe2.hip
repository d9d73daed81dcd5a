// probe2 -- which instruction class of the median kernel goes wrong beside a dense int8-MFMA stream of another stream?
// The victim classes are run alone (reference) and again under the co-runner; outputs are compared word for word.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hunt/probe2.hip -o tools/hunt/probe2.bin -Lvip_amd -lvipmi -Wl,-rpath,$PWD/vip_amd
#include "../../vip_amd/csrc/collapse.hip"
#include "aggressors.hip"
#include <thread>
#include <atomic>
#include <vector>
#include <cstdlib>
using namespace vipmi;

__device__ __forceinline__ unsigned hash3(unsigned a, unsigned b, unsigned c) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}
__device__ __forceinline__ float rndf(unsigned h) { return ((float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 6.0f; }

constexpr int RPL = 7;
namespace vipmi { namespace {
#include "median_keys_dbg.inc"
} }
template <int CLS>
__device__ __forceinline__ void victim_body(int iters, unsigned* __restrict__ out, const unsigned* __restrict__ gsrc, unsigned* __restrict__ out2) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned* hist = reinterpret_cast<unsigned*>(smem) + HIST_WORDS * wave;
  for (int it = 0; it < iters; ++it) {
    unsigned key[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const unsigned h = hash3(blockIdx.x * 512u + threadIdx.x, (unsigned)it, (unsigned)r);
      key[r] = (lane + 64 * r < 400) ? f2key(rndf(h)) : 0xffffffffu;
    }
    unsigned res = 0;
    if (CLS == 0) {                                 // per-lane counts summed with ballots
      const unsigned cand = f2key(rndf(hash3(blockIdx.x, wave, it)));
      int c = 0;
#pragma unroll
      for (int r = 0; r < RPL; ++r) c += key[r] < cand ? 1 : 0;
      res = (unsigned)wave_count<RPL>(c);
    } else if (CLS == 1) {                          // DPP butterflies + readlane: wave minimum / maximum
      unsigned lo = 0xffffffffu, hi = 0;
#pragma unroll
      for (int r = 0; r < RPL; ++r) { lo = umin_(lo, key[r]); hi = umax_(hi, key[r] == 0xffffffffu ? 0u : key[r]); }
      res = wave_min_u32(lo) ^ (wave_max_u32(hi) >> 1);
    } else if (CLS == 2) {                          // DPP prefix sum
      res = wave_inclusive_sum(key[0] & 7u);
    } else if (CLS == 3) {                          // readlane broadcast loop
      const unsigned cand = key[1];
      int less = 0;
      for (unsigned q = 0; q < 64; ++q) {
        const unsigned kq = (unsigned)__builtin_amdgcn_readlane((int)cand, (int)q);
        less += (kq < cand) ? 1 : 0;
      }
      res = (unsigned)less;
    } else if (CLS == 4) {                          // LDS histogram with returning atomics
      reinterpret_cast<uint4*>(hist)[lane] = make_uint4(0u, 0u, 0u, 0u);
      if (lane == 0) hist[256] = 0u;
      wave_lds_sync();
      unsigned ordsum = 0;
#pragma unroll
      for (int r = 0; r < RPL; ++r) ordsum += atomicAdd(&hist[key[r] == 0xffffffffu ? 256 : (key[r] >> 9) & 255], 1u) * (r + 1);
      wave_lds_sync();
      const uint4 h = reinterpret_cast<const uint4*>(hist)[lane];
      // ordinals depend on the lane order of the atomic unit: only their per-bin SET is fixed -> output the counts only
      res = h.x + 3 * h.y + 5 * h.z + 7 * h.w;
      (void)ordsum;
      wave_lds_sync();
    } else if (CLS == 5) {                          // LDS slot stores then reads
#pragma unroll
      for (int r = 0; r < RPL; ++r) hist[(lane * 5 + r * 64 + it) % 384] = key[r];    // 5 is odd: a permutation of 0..383 within r-planes up to overlaps
      wave_lds_sync();
      res = hist[lane] ^ hist[lane + 64] ^ hist[lane + 128] ^ hist[lane + 320];
      wave_lds_sync();
    } else if (CLS == 6) {                          // float binning (division, conversions)
      unsigned lo = 0xffffffffu, hi = 0;
#pragma unroll
      for (int r = 0; r < RPL; ++r) { lo = umin_(lo, key[r]); hi = umax_(hi, key[r] == 0xffffffffu ? 0u : key[r]); }
      const float flo = key2f(lo), fhi = key2f(hi);
      const float scale = 256.0f / (fhi - flo);
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        int b = (int)((key2f(key[r]) - flo) * scale);
        b = b > 255 ? 255 : b;
        res += (key[r] >= lo && key[r] <= hi) ? (unsigned)b * (r + 1) : 256u;
      }
    } else if (CLS == 7) {                          // the whole selection
      unsigned klow, khigh;
      median_keys<RPL>(key, 400, hist, lane, klow, khigh);
      res = klow ^ (khigh * 3u);
    } else if (CLS >= 11 && CLS <= 17) {            // the selection, stopped after stage CLS - 10
      unsigned klow = 0, khigh = 0, dbg = 0xdead0000u;
      median_keys_dbg<RPL, CLS - 10>(key, 400, hist, lane, klow, khigh, dbg);
      res = dbg;
      if (CLS == 13 && out2) { out2[(((size_t)blockIdx.x * iters + it) * 512 + threadIdx.x) * 2] = klow; out2[(((size_t)blockIdx.x * iters + it) * 512 + threadIdx.x) * 2 + 1] = khigh; }
    } else if (CLS == 8) {                          // bisection
      res = select_rank<RPL>(key, 199);
    } else if (CLS == 9) {                          // plain VALU
#pragma unroll
      for (int r = 0; r < RPL; ++r) res = res * 31u + (key[r] ^ (key[r] >> 7)) + (res >> 3);
    } else if (CLS == 10) {                         // global loads
      for (int r = 0; r < RPL; ++r) res += gsrc[(hash3(blockIdx.x, threadIdx.x, it * 8 + r) & 0xFFFFFFu)];
    }
    out[((size_t)blockIdx.x * iters + it) * 512 + threadIdx.x] = res;
  }
}

// the classes as the compiler likes them (packed FP32 allowed) ...
template <int CLS>
__global__ __launch_bounds__(512) void victim(int iters, unsigned* __restrict__ out, const unsigned* __restrict__ gsrc, unsigned* __restrict__ out2) {
  victim_body<CLS>(iters, out, gsrc, out2);
}
// ... and as the library now builds its median kernels: no packed-FP32 selection (common.h VIPMI_NO_PK32)
template <int CLS>
__global__ VIPMI_NO_PK32 __launch_bounds__(512) void victim_nopk(int iters, unsigned* __restrict__ out, const unsigned* __restrict__ gsrc, unsigned* __restrict__ out2) {
  victim_body<CLS>(iters, out, gsrc, out2);
}
typedef void (*vk_t)(int, unsigned*, const unsigned*, unsigned*);
int main(int argc, char** argv) {
  const int blocks = 2048, iters = argc > 1 ? atoi(argv[1]) : 16, reps = 4;
  const int agg_kind = argc > 2 ? atoi(argv[2]) : 2, agg_lds = argc > 3 ? atoi(argv[3]) : 81920, agg_iters = argc > 4 ? atoi(argv[4]) : 100;
  const int vict_lds = argc > 5 ? atoi(argv[5]) : 39488;
  const size_t words = (size_t)blocks * iters * 512;
  unsigned *out, *ref, *gsrc, *sink, *out2, *ref2;
  hipMalloc(&out2, (size_t)2048 * 16 * 512 * 8); hipMalloc(&ref2, (size_t)2048 * 16 * 512 * 8);
  hipMalloc(&out, words * 4); hipMalloc(&ref, words * 4); hipMalloc(&gsrc, (size_t)64 << 20); hipMalloc(&sink, 64);
  {
    std::vector<unsigned> h((size_t)16 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
    hipMemcpy(gsrc, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  }
  hipStream_t s1, s2;
  hipStreamCreate(&s1); hipStreamCreate(&s2);
  vk_t ks[] = {victim<0>, victim<1>, victim<2>, victim<3>, victim<4>, victim<5>, victim<6>, victim<7>, victim<8>, victim<9>, victim<10>, victim<11>, victim<13>, victim<13>, victim<14>, victim<15>, victim<16>, victim<17>, victim_nopk<7>, victim_nopk<13>};
  const char* names[] = {"ballot counts", "DPP min/max + readlane", "DPP prefix sum", "readlane loop", "LDS atomics histogram", "LDS slot store/read",
                         "float binning", "median_keys", "bisection", "plain VALU", "global loads", "sel: lo/hi", "sel: bins+scale", "sel: histogram", "sel: prefix sums", "sel: L/bstar/c/rem", "sel: candidates (sum)", "sel: less", "median_keys, NO_PK32", "sel: histogram, NO_PK32"};
  std::vector<unsigned> ho(words), hr(words);
  for (int c = (argc > 6 ? atoi(argv[6]) : 0); c < 20; ++c) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(ks[c]), hipFuncAttributeMaxDynamicSharedMemorySize, vict_lds);
    hipLaunchKernelGGL(ks[c], dim3(blocks), dim3(512), vict_lds, s1, iters, ref, gsrc, c == 13 ? ref2 : nullptr);
    hipStreamSynchronize(s1);
    hipMemcpy(hr.data(), ref, words * 4, hipMemcpyDeviceToHost);
    // a second solo run must agree (determinism of the class itself)
    hipLaunchKernelGGL(ks[c], dim3(blocks), dim3(512), vict_lds, s1, iters, out, gsrc, c == 13 ? out2 : nullptr);
    hipStreamSynchronize(s1);
    hipMemcpy(ho.data(), out, words * 4, hipMemcpyDeviceToHost);
    size_t solo = 0;
    for (size_t i = 0; i < words; ++i) solo += ho[i] != hr[i];
    std::atomic<bool> stop{false};
    std::atomic<int> nag{0};
    std::thread ag([&] {
      hipSetDevice(0);
      while (!stop) { vipmi_hunt_aggressor(s2, agg_kind, agg_lds, 4096, agg_iters, gsrc, (size_t)64 << 20, sink); hipStreamSynchronize(s2); nag++; }
    });
    while (nag < 2) std::this_thread::yield();
    size_t bad[4] = {0, 0, 0, 0}, badwaves = 0, first = (size_t)-1;
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(ks[c], dim3(blocks), dim3(512), vict_lds, s1, iters, out, gsrc, c == 13 ? out2 : nullptr);
      hipStreamSynchronize(s1);
      hipMemcpy(ho.data(), out, words * 4, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < words; ++i)
        if (ho[i] != hr[i]) { ++bad[r]; if (first == (size_t)-1) first = i; }
      if (r == 0 && c == 13) {
        std::vector<unsigned> a2(words * 2), r2(words * 2);
        hipMemcpy(a2.data(), out2, words * 8, hipMemcpyDeviceToHost);
        hipMemcpy(r2.data(), ref2, words * 8, hipMemcpyDeviceToHost);
        size_t wb = 0, wbins = 0, wscale = 0, whist = 0, shown = 0;
        for (size_t w = 0; w < words / 64; ++w) {
          bool bh = false, bb = false, bs = false;
          int nbl = 0;
          for (int l = 0; l < 64; ++l) {
            const size_t i = w * 64 + l;
            bh |= ho[i] != hr[i];
            if (a2[2 * i] != r2[2 * i]) { bb = true; ++nbl; }
            bs |= a2[2 * i + 1] != r2[2 * i + 1];
          }
          if (bh || bb || bs) {
            ++wb; whist += bh; wbins += bb; wscale += bs;
            if (shown < 12) {
              ++shown;
              printf("   wave-iter %zu (block %zu it %zu wave %zu): hist %d bins %d (%d lanes) scale/lo/hi %d", w, w / (8 * 16), (w / 8) % 16, w % 8, bh, bb, nbl, bs);
              for (int l = 0; l < 64; ++l) { const size_t i = w * 64 + l; if (a2[2 * i] != r2[2 * i] && nbl <= 70) { printf("  lane %d bins %08x want %08x", l, a2[2 * i], r2[2 * i]); break; } }
              if (bs) printf("  scale^lo^hi got %08x want %08x", a2[2 * w * 64 + 1], r2[2 * w * 64 + 1]);
              printf("\n");
            }
          }
        }
        printf("   class 13 detail: wave-iterations wrong %zu: hist differs %zu, bins differ %zu, scale/lo/hi differ %zu\n", wb, whist, wbins, wscale);
      }
      if (r == 0)
        for (size_t w = 0; w < words / 64; ++w) { bool b = false; for (int l = 0; l < 64; ++l) b |= ho[w * 64 + l] != hr[w * 64 + l]; badwaves += b; }
    }
    stop = true; ag.join();
    printf("class %2d %-26s solo-mismatch %zu | under co-runner: wrong words %zu %zu %zu %zu of %zu (wave-iterations touched in run 0: %zu)", c, names[c], solo, bad[0], bad[1], bad[2],
           bad[3], words, badwaves);
    if (first != (size_t)-1) printf("  first: word %zu got %08x want %08x", first, ho[first], hr[first]);
    printf("  [co-runner launches %d]\n", nag.load());
    fflush(stdout);
  }
  return 0;
}
