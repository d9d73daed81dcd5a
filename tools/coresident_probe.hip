// A probe workgroup that shares CUs with the int8 Gram product: it fills its LDS and a set of registers with patterns and
// re-checks them for a while; reports whether LDS or registers were changed under it.
//   hipcc --offload-arch=gfx950 -O2 tools/coresident_probe.hip -o tools/coresident_probe.bin -ldl ; tools/coresident_probe.bin vip_amd/libvipmi.so [lds_kb] [gram_i8]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <thread>
#include <atomic>
#include <vector>
struct vipmi_ctx;
typedef int (*create_t)(int, void*, vipmi_ctx**);
typedef int (*gram_t)(vipmi_ctx*, const float*, int64_t, int64_t, int64_t, double*);
typedef int (*setopt_t)(vipmi_ctx*, const char*, int64_t);

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_rows(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false); }
__device__ __forceinline__ unsigned umin_(unsigned a, unsigned b) { return a < b ? a : b; }
__global__ __launch_bounds__(512) void probe(int words, int iters, unsigned* bad_lds, unsigned* bad_reg, unsigned* first, const unsigned* gbuf, size_t gwords,
                                              unsigned* bad_glob, unsigned* bad_atom, unsigned* bad_dpp) {
  extern __shared__ unsigned sm[];
  const unsigned tag = 0x5a000000u ^ (blockIdx.x * 7919u);
  for (int e = threadIdx.x; e < words; e += blockDim.x) sm[e] = tag + e;
  unsigned r[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) r[i] = tag * 31u + threadIdx.x * 24 + i;
  __syncthreads();
  unsigned nl = 0, nr = 0;
  for (int it = 0; it < iters; ++it) {
    for (int e = threadIdx.x; e < words; e += blockDim.x) {
      const unsigned v = sm[e];
      if (v != tag + e) { ++nl; if (atomicAdd(first, 1u) < 8) printf("block %d LDS word %d: %08x expected %08x\n", blockIdx.x, e, v, tag + e); sm[e] = tag + e; }
    }
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      asm volatile("" : "+v"(r[i]));
      if (r[i] != tag * 31u + threadIdx.x * 24 + i) { ++nr; r[i] = tag * 31u + threadIdx.x * 24 + i; }
    }
    // global loads of a known pattern (gbuf[i] = i * 2654435761u), 16 bytes per lane, rows 1 MB apart like the median's tile load
    {
      unsigned ng = 0;
      for (int k = 0; k < 8; ++k) {
        const size_t i = ((size_t)(blockIdx.x * 64 + (it * 8 + k) % 400) * 262144 + (size_t)threadIdx.x * 4) % (gwords - 4);
        const size_t i4 = i & ~(size_t)3;
        const uint4 v = *reinterpret_cast<const uint4*>(gbuf + i4);
        if (v.x != (unsigned)(i4 * 2654435761u) || v.w != (unsigned)((i4 + 3) * 2654435761u)) ++ng;
      }
      if (ng) atomicAdd(bad_glob, ng);
    }
    // LDS atomics: every lane adds 1 to one of 64 counters of its wave, then the counters must sum to 64
    {
      __syncthreads();
      unsigned* cnt = sm + words + (threadIdx.x >> 6) * 64;
      cnt[threadIdx.x & 63] = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      atomicAdd(&cnt[(threadIdx.x * 7 + it) & 63], 1u);
      atomicAdd(&cnt[(threadIdx.x * 13 + it) & 15], 1u);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      unsigned v = cnt[threadIdx.x & 63];
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      if (v != 128u && (threadIdx.x & 63) == 0) atomicAdd(bad_atom, 1u);
      __syncthreads();
    }
    // DPP butterflies + readlane (wave minimum), DPP prefix sum, ballots -- against values known in closed form
    {
      const unsigned lane = threadIdx.x & 63;
      unsigned v = (lane * 2654435761u + it * 97u) | 1u;
      unsigned m = v;
      m = umin_(m, dpp_u32<0xB1>(m)); m = umin_(m, dpp_u32<0x4E>(m)); m = umin_(m, dpp_u32<0x141>(m)); m = umin_(m, dpp_u32<0x140>(m));
      const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)m, 0), b = (unsigned)__builtin_amdgcn_readlane((int)m, 16);
      const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)m, 32), d = (unsigned)__builtin_amdgcn_readlane((int)m, 48);
      const unsigned wmin = umin_(umin_(a, b), umin_(c, d));
      unsigned ref = 0xffffffffu;
      for (unsigned l = 0; l < 64; ++l) ref = umin_(ref, (l * 2654435761u + it * 97u) | 1u);
      unsigned p = lane & 3;                         // inclusive prefix sum of lane & 3
      p += dpp_rows<0x111, 0xf>(p); p += dpp_rows<0x112, 0xf>(p); p += dpp_rows<0x114, 0xf>(p); p += dpp_rows<0x118, 0xf>(p);
      p += dpp_rows<0x142, 0xa>(p); p += dpp_rows<0x143, 0xc>(p);
      unsigned pref = 0;
      for (unsigned l = 0; l <= lane; ++l) pref += l & 3;
      const unsigned long long bal = __ballot((lane * 5 + it) % 3 == 0);
      unsigned long long bref = 0;
      for (unsigned l = 0; l < 64; ++l) if ((l * 5 + it) % 3 == 0) bref |= 1ull << l;
      if (wmin != ref || p != pref || bal != bref) atomicAdd(bad_dpp, 1u);
    }
    __builtin_amdgcn_s_sleep(20);
  }
  if (nl) atomicAdd(bad_lds, nl);
  if (nr) atomicAdd(bad_reg, nr);
}

int main(int argc, char** argv) {
  void* h = dlopen(argv[1], RTLD_NOW);
  if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
  create_t create = (create_t)dlsym(h, "vipmi_create");
  gram_t gram = (gram_t)dlsym(h, "vipmi_gram_f32");
  setopt_t setopt = (setopt_t)dlsym(h, "vipmi_set_option");
  const int lds_kb = argc > 2 ? atoi(argv[2]) : 39;
  const int i8 = argc > 3 ? atoi(argv[3]) : 1;
  const int64_t n = 400, P = 262144;
  float* M; double* G;
  hipMalloc(&M, n * P * 4); hipMalloc(&G, n * n * 8);
  std::vector<float> hm(n * P);
  for (size_t i = 0; i < hm.size(); ++i) hm[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(M, hm.data(), n * P * 4, hipMemcpyHostToDevice);
  hipStream_t s1, s2;
  hipStreamCreate(&s1); hipStreamCreate(&s2);
  vipmi_ctx* ctx = nullptr;
  int rc = create(0, s1, &ctx);
  int rc2 = setopt(ctx, "gram_i8", i8);
  printf("create %d setopt %d\n", rc, rc2);
  std::atomic<bool> stop{false};
  std::atomic<int> ngram{0}, gerr{0};
  std::thread loader([&] { hipSetDevice(0); while (!stop) { if (gram(ctx, M, n, P, P, G)) gerr++; hipStreamSynchronize(s1); ngram++; } });
  unsigned *bl, *br, *fi, *bg, *ba, *gbuf, *bd; hipMalloc(&bd, 4); hipMemset(bd, 0, 4);
  hipMalloc(&bl, 4); hipMalloc(&br, 4); hipMalloc(&fi, 4); hipMalloc(&bg, 4); hipMalloc(&ba, 4);
  hipMemset(bl, 0, 4); hipMemset(br, 0, 4); hipMemset(fi, 0, 4); hipMemset(bg, 0, 4); hipMemset(ba, 0, 4);
  const size_t gwords = (size_t)100 << 20;
  hipMalloc(&gbuf, gwords * 4);
  { std::vector<unsigned> hg(gwords); for (size_t i = 0; i < gwords; ++i) hg[i] = (unsigned)(i * 2654435761u); hipMemcpy(gbuf, hg.data(), gwords * 4, hipMemcpyHostToDevice); }
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
  for (int rep = 0; rep < 20; ++rep) {
    hipLaunchKernelGGL(probe, dim3(1024), dim3(512), lds_kb * 1024, s2, lds_kb * 256 - 512, argc > 4 ? atoi(argv[4]) : 1500, bl, br, fi, gbuf, gwords, bg, ba, bd);
    hipStreamSynchronize(s2);
  }
  stop = true; loader.join();
  printf("gram calls %d (errors %d)\n", ngram.load(), gerr.load());
  unsigned a, b, c, d, e5; hipMemcpy(&e5, bd, 4, hipMemcpyDeviceToHost);
  hipMemcpy(&a, bl, 4, hipMemcpyDeviceToHost); hipMemcpy(&b, br, 4, hipMemcpyDeviceToHost); hipMemcpy(&c, bg, 4, hipMemcpyDeviceToHost); hipMemcpy(&d, ba, 4, hipMemcpyDeviceToHost);
  printf("probe LDS %d KB per workgroup, gram_i8=%d: corrupted LDS words %u, registers %u, wrong global loads %u, wrong LDS-atomic sums %u, wrong DPP / readlane / ballot results %u\n", lds_kb, i8, a, b, c, d, e5);
  return 0;
}
