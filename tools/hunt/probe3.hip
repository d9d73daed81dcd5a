// probe3 -- does a transcendental (TRANS-pipe) result reach its consumer too late when ANOTHER wave of the CU streams MFMAs?
// The compiler separates v_rcp_f32 and its first use by ONE wait state (`s_nop 0`, gfx940 "VALUTransUseHazard").  Here the
// sequence is written in inline asm with N wait states; each class runs alone (reference) and under a co-runner of another stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hunt/probe3.hip -o tools/hunt/probe3.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <atomic>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef double v4d __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned hash3(unsigned a, unsigned b, unsigned c) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

#define STR2(x) #x
#define STR(x) STR2(x)
// OP dst, src ; s_nop N (N >= 0) or nothing (N < 0) ; v_mul_f32 dst, dst, y
#define TRANS_SEQ(OP, NOPSTR) asm volatile(OP " %0, %1\n\t" NOPSTR "v_mul_f32 %0, %0, %2" : "=&v"(r) : "v"(x), "v"(y))

template <int CLS>
__global__ __launch_bounds__(512) void victim(int iters, unsigned* __restrict__ out) {
  extern __shared__ float smem[];
  if (threadIdx.x == 0) smem[0] = 1.f;
  for (int it = 0; it < iters; ++it) {
    const unsigned h = hash3(blockIdx.x * 512u + threadIdx.x, (unsigned)it, 77u);
    const float x = 1.0f + (float)(h >> 8) * (1.0f / 16777216.0f) * 6.0f, y = 3.0f + (float)(h & 255u);
    float r;
    // poison the destination first so that a stale read is visible: r = y (a different value every iteration)
    if (CLS == 0) TRANS_SEQ("v_rcp_f32", "s_nop 0\n\t");
    else if (CLS == 1) TRANS_SEQ("v_rcp_f32", "s_nop 1\n\t");
    else if (CLS == 2) TRANS_SEQ("v_rcp_f32", "s_nop 2\n\t");
    else if (CLS == 3) TRANS_SEQ("v_rcp_f32", "s_nop 3\n\t");
    else if (CLS == 4) TRANS_SEQ("v_rcp_f32", "s_nop 5\n\t");
    else if (CLS == 5) TRANS_SEQ("v_rcp_f32", "s_nop 7\n\t");
    else if (CLS == 6) TRANS_SEQ("v_rcp_f32", "s_nop 7\n\ts_nop 7\n\t");
    else if (CLS == 7) TRANS_SEQ("v_sqrt_f32", "s_nop 0\n\t");
    else if (CLS == 8) TRANS_SEQ("v_exp_f32", "s_nop 0\n\t");
    else if (CLS == 9) TRANS_SEQ("v_rsq_f32", "s_nop 0\n\t");
    else if (CLS == 10) TRANS_SEQ("v_log_f32", "s_nop 0\n\t");
    else if (CLS == 11) TRANS_SEQ("v_cvt_i32_f32", "s_nop 0\n\t");       // not a TRANS op: control
    else if (CLS == 12) TRANS_SEQ("v_rcp_f32", "v_mov_b32 %0, %0\n\t");  // a dependent v_mov right behind (no nop at all)
    else if (CLS == 13) { r = 256.0f / x * y; }                           // compiler's own division
    else if (CLS == 14) { r = __builtin_amdgcn_rcpf(x) * y; }             // compiler's own rcp + use
    else if (CLS == 15) { r = __builtin_sqrtf(x) * y; }
    else if (CLS == 17) {                                                  // packed multiply, both halves
      float2 a = make_float2(x, y), b = make_float2(y, x), c;
      asm volatile("v_pk_mul_f32 %0, %1, %2\n\ts_nop 4" : "=&v"(c) : "v"(a), "v"(b));
      r = c.x + c.y;
    } else if (CLS == 18) {                                                // packed multiply by a broadcast low half (op_sel_hi:[1,0])
      float2 a = make_float2(x, y), b = make_float2(y, x), c;
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\ts_nop 4" : "=&v"(c) : "v"(a), "v"(b));
      r = c.x + c.y;
    } else if (CLS == 19) {                                                // packed subtract of a broadcast high half
      float2 a = make_float2(x, y), b = make_float2(y, x), c;
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 4" : "=&v"(c) : "v"(a), "v"(b));
      r = c.x + c.y;
    } else if (CLS == 20) {                                                // the compiler's chain, back to back, cvt right behind
      float2 a = make_float2(x, y), b = make_float2(y, x), s = make_float2(1.5f, 0.f);
      asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 %0, %0, %2 op_sel_hi:[1,0]" : "+v"(a) : "v"(b), "v"(s));
      r = (float)((int)a.x + 3 * (int)a.y);
    } else if (CLS == 21) {
      float2 a = make_float2(x, y), b = make_float2(y, x), c = make_float2(0.5f, 0.25f), d;
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3\n\ts_nop 4" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
      r = d.x + d.y;
    } else if (CLS == 22) {                                                // plain C++: what the median's binning is compiled from
      const float flo = 1.0f, scale = 256.0f / 6.0f;
      const float k0 = x < y ? x : 1.f + x * 0.5f, k1 = 1.0f + (y - 3.0f) * (6.0f / 255.0f);
      int b0 = (int)((k0 - flo) * scale), b1 = (int)((k1 - flo) * scale);
      b0 = b0 > 255 ? 255 : b0; b1 = b1 > 255 ? 255 : b1;
      r = (float)(b0 * 257 + b1);
    }
    else { r = x * y; }
    out[((size_t)blockIdx.x * iters + it) * 512 + threadIdx.x] = __float_as_uint(r);
  }
}

template <int KIND>
__global__ __launch_bounds__(256, 2) void corunner(int iters, unsigned* __restrict__ sink) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x;
  if (tid == 0) smem[0] = 1.f;
  unsigned acc = 0;
  if (KIND == 0) {                                  // v_mfma_i32_16x16x64_i8
    v4i a = {tid, 1, 2, 3}, b = {4, tid, 6, 7}, c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, b, c3, 0, 0, 0);
      }
    acc = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (KIND == 1) {                           // v_mfma_f64_16x16x4_f64
    double a = tid, b = 1.5; v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
      }
    acc = (unsigned)(c0[0] + c1[1] + c2[2] + c3[3]);
  } else if (KIND == 2) {                           // v_mfma_f32_32x32x2_f32
    float a = tid, b = 1.5f; v16f c0 = {}, c1 = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
      }
    acc = (unsigned)(c0[0] + c1[1]);
  } else if (KIND == 3) {                           // v_mfma_f32_16x16x32_bf16 (gfx950)
    v8s a = {1, 2, 3, 4, 5, 6, 7, (short)tid}, b = {8, 7, 6, 5, 4, 3, 2, 1}; v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16((__bf16 __attribute__((ext_vector_type(8))))a, (__bf16 __attribute__((ext_vector_type(8))))b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16((__bf16 __attribute__((ext_vector_type(8))))b, (__bf16 __attribute__((ext_vector_type(8))))a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16((__bf16 __attribute__((ext_vector_type(8))))a, (__bf16 __attribute__((ext_vector_type(8))))a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16((__bf16 __attribute__((ext_vector_type(8))))b, (__bf16 __attribute__((ext_vector_type(8))))b, c3, 0, 0, 0);
      }
    acc = (unsigned)(c0[0] + c1[1] + c2[2] + c3[3]);
  } else if (KIND == 4) {                           // v_mfma_i32_32x32x32_i8 (gfx950)
    typedef int v16i __attribute__((ext_vector_type(16)));
    v4i a = {tid, 1, 2, 3}, b = {4, tid, 6, 7}; v16i c0 = {}, c1 = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, c1, 0, 0, 0);
      }
    acc = c0[0] + c1[1];
  } else if (KIND == 5) {                           // plain VALU stream (packed FP32)
    float a = tid, b = 1.0001f;
    for (int it = 0; it < iters * 16; ++it) { a = a * b + 0.5f; b = b * 0.9999f + 1e-6f; }
    acc = __float_as_uint(a + b);
  } else if (KIND == 6) {                           // TRANS stream of the co-runner itself
    float a = tid + 1.f;
    for (int it = 0; it < iters * 8; ++it) a = __builtin_amdgcn_rcpf(a) + 1.5f;
    acc = __float_as_uint(a);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

typedef void (*vk_t)(int, unsigned*);
typedef void (*ck_t)(int, unsigned*);
int main(int argc, char** argv) {
  const int blocks = 2048, iters = 16, reps = 3;
  const int vict_lds = argc > 1 ? atoi(argv[1]) : 39488, co_lds = argc > 2 ? atoi(argv[2]) : 81920;
  const size_t words = (size_t)blocks * iters * 512;
  unsigned *out, *sink;
  hipMalloc(&out, words * 4); hipMalloc(&sink, 64);
  hipStream_t s1, s2;
  hipStreamCreate(&s1); hipStreamCreate(&s2);
  vk_t vs[] = {victim<0>, victim<1>, victim<2>, victim<3>, victim<4>, victim<5>, victim<6>, victim<7>, victim<8>, victim<9>, victim<10>, victim<11>,
               victim<12>, victim<13>, victim<14>, victim<15>, victim<16>, victim<17>, victim<18>, victim<19>, victim<20>, victim<21>, victim<22>};
  const char* vn[] = {"rcp nop0", "rcp nop1", "rcp nop2", "rcp nop3", "rcp nop5", "rcp nop7", "rcp nop7+7", "sqrt nop0", "exp nop0", "rsq nop0", "log nop0",
                      "cvt nop0 (ctl)", "rcp, v_mov", "256/x (compiler)", "rcpf*y (compiler)", "sqrtf*y (compiler)", "x*y (ctl)", "pk_mul", "pk_mul op_sel_hi", "pk_add op_sel neg", "pk chain + cvt", "pk_fma", "cndmask -> pk chain"};
  ck_t cs[] = {corunner<0>, corunner<1>, corunner<2>, corunner<3>, corunner<4>, corunner<5>, corunner<6>};
  const char* cn[] = {"i8 16x16x64", "f64 16x16x4", "f32 32x32x2", "bf16 16x16x32", "i8 32x32x32", "VALU fma", "TRANS rcp"};
  const int nv = 23, nc = 7;
  const int v0 = argc > 3 ? atoi(argv[3]) : 0;
  std::vector<std::vector<unsigned>> ref(nv, std::vector<unsigned>(words));
  std::vector<unsigned> ho(words);
  for (int v = 0; v < nv; ++v) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(vs[v]), hipFuncAttributeMaxDynamicSharedMemorySize, vict_lds);
    hipLaunchKernelGGL(vs[v], dim3(blocks), dim3(512), vict_lds, s1, iters, out);
    hipStreamSynchronize(s1);
    hipMemcpy(ref[v].data(), out, words * 4, hipMemcpyDeviceToHost);
  }
  printf("%-20s", "victim \\ co-runner");
  for (int c = 0; c < nc; ++c) printf(" %14s", cn[c]);
  printf("\n");
  for (int v = v0; v < nv; ++v) {
    printf("%-20s", vn[v]);
    size_t lanehist[64] = {0};
    for (int c = 0; c < nc; ++c) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(cs[c]), hipFuncAttributeMaxDynamicSharedMemorySize, co_lds);
      std::atomic<bool> stop{false};
      std::atomic<int> nag{0};
      std::thread ag([&] {
        hipSetDevice(0);
        while (!stop) { hipLaunchKernelGGL(cs[c], dim3(4096), dim3(256), co_lds, s2, 100, sink); hipStreamSynchronize(s2); nag++; }
      });
      while (nag < 2) std::this_thread::yield();
      size_t bad = 0;
      for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(vs[v], dim3(blocks), dim3(512), vict_lds, s1, iters, out);
        hipStreamSynchronize(s1);
        hipMemcpy(ho.data(), out, words * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < words; ++i) if (ho[i] != ref[v][i]) { ++bad; ++lanehist[i & 63]; }
      }
      stop = true; ag.join();
      printf(" %14zu", bad);
      fflush(stdout);
    }
    printf("\n");
    size_t tot = 0; for (int l = 0; l < 64; ++l) tot += lanehist[l];
    if (tot) { printf("      wrong words by lane quarter: %zu %zu %zu %zu\n", lanehist[0] + lanehist[5] * 0, 0ul, 0ul, 0ul);
      size_t q[4] = {0, 0, 0, 0}; for (int l = 0; l < 64; ++l) q[l >> 4] += lanehist[l]; printf("      lanes 0-15: %zu, 16-31: %zu, 32-47: %zu, 48-63: %zu\n", q[0], q[1], q[2], q[3]); }
  }
  printf("(wrong words of %zu per cell, %d runs each)\n", words * reps, reps);
  return 0;
}
