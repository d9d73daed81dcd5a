// probe4 -- which packed-FP32 forms return wrong lanes while ANOTHER wave of the CU streams MFMAs?  (found with probe2 / probe3:
// `v_pk_add_f32 ... op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]` gives wrong results in lanes 48..63 beside v_mfma_i32_16x16x64_i8.)
// Every (instruction, modifier) form runs alone (reference) and under each co-runner; outputs are compared word for word and the
// wrong lanes are shown with their inputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hunt/probe4.hip -o tools/hunt/probe4.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <atomic>
#include <utility>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef short v8s __attribute__((ext_vector_type(8)));

__host__ __device__ inline unsigned hash3(unsigned a, unsigned b, unsigned c) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}
__host__ __device__ inline void inputs(unsigned gtid, unsigned it, float& ax, float& ay, float& bx, float& by) {
  const unsigned h = hash3(gtid, it, 77u);
  ax = 1.0f + (float)(h >> 8) * (1.0f / 16777216.0f) * 6.0f;
  ay = 3.0f + (float)(h & 255u);
  bx = 100.0f + (float)((h >> 4) & 1023u);
  by = 0.25f + (float)((h >> 14) & 63u) * 0.125f;
}

#define PK2(OPSTR, MODSTR) asm volatile(OPSTR " %0, %1, %2 " MODSTR "\n\ts_nop 4" : "=&v"(c) : "v"(a), "v"(b))
#define PK3(MODSTR) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 " MODSTR "\n\ts_nop 4" : "=&v"(c) : "v"(a), "v"(b), "v"(d))

template <int F>
__global__ __launch_bounds__(512) void victim(int iters, float2* __restrict__ out) {
  extern __shared__ float smem[];
  if (threadIdx.x == 0) smem[0] = 1.f;
  for (int it = 0; it < iters; ++it) {
    float2 a, b, c, d = make_float2(0.5f, 0.25f);
    inputs(blockIdx.x * 512u + threadIdx.x, (unsigned)it, a.x, a.y, b.x, b.y);
    if (F == 0) PK2("v_pk_add_f32", "");
    else if (F == 1) PK2("v_pk_add_f32", "op_sel:[0,1]");
    else if (F == 2) PK2("v_pk_add_f32", "op_sel:[1,0]");
    else if (F == 3) PK2("v_pk_add_f32", "op_sel:[1,1]");
    else if (F == 4) PK2("v_pk_add_f32", "op_sel_hi:[0,1]");
    else if (F == 5) PK2("v_pk_add_f32", "op_sel_hi:[1,0]");
    else if (F == 6) PK2("v_pk_add_f32", "op_sel_hi:[0,0]");
    else if (F == 7) PK2("v_pk_add_f32", "neg_lo:[0,1] neg_hi:[0,1]");
    else if (F == 8) PK2("v_pk_add_f32", "op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]");
    else if (F == 9) PK2("v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0]");
    else if (F == 10) PK2("v_pk_mul_f32", "");
    else if (F == 11) PK2("v_pk_mul_f32", "op_sel:[0,1]");
    else if (F == 12) PK2("v_pk_mul_f32", "op_sel:[1,0]");
    else if (F == 13) PK2("v_pk_mul_f32", "op_sel:[1,1]");
    else if (F == 14) PK2("v_pk_mul_f32", "op_sel_hi:[0,1]");
    else if (F == 15) PK2("v_pk_mul_f32", "op_sel_hi:[1,0]");
    else if (F == 16) PK2("v_pk_mul_f32", "op_sel_hi:[0,0]");
    else if (F == 17) PK2("v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[1,0]");
    else if (F == 18) PK3("");
    else if (F == 19) PK3("op_sel:[0,1,0]");
    else if (F == 20) PK3("op_sel:[1,0,0]");
    else if (F == 21) PK3("op_sel:[0,0,1]");
    else if (F == 22) PK3("op_sel_hi:[0,1,1]");
    else if (F == 23) PK3("op_sel_hi:[1,0,1]");
    else if (F == 24) PK3("op_sel_hi:[1,1,0]");
    else if (F == 25) PK3("op_sel:[0,1,0] op_sel_hi:[1,0,1]");
    else if (F == 26) PK3("op_sel:[1,1,0] op_sel_hi:[0,0,1]");
    else if (F == 27) PK3("op_sel:[0,1,0] neg_lo:[0,1,0] neg_hi:[0,1,0]");
    else if (F == 28) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]\n\ts_nop 4" : "=&v"(c) : "v"(a), "v"(b)); }
    else if (F == 29) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]\n\ts_nop 4" : "=&v"(c) : "v"(a), "v"(b)); }
    else { c = make_float2(a.x + b.y, a.y + b.y); }
    out[((size_t)blockIdx.x * iters + it) * 512 + threadIdx.x] = c;
  }
}
static const char* FN[] = {"pk_add", "pk_add op_sel:[0,1]", "pk_add op_sel:[1,0]", "pk_add op_sel:[1,1]", "pk_add op_sel_hi:[0,1]", "pk_add op_sel_hi:[1,0]",
                           "pk_add op_sel_hi:[0,0]", "pk_add neg src1", "pk_add op_sel:[0,1] neg src1", "pk_add op_sel:[0,1] hi:[1,0]",
                           "pk_mul", "pk_mul op_sel:[0,1]", "pk_mul op_sel:[1,0]", "pk_mul op_sel:[1,1]", "pk_mul op_sel_hi:[0,1]", "pk_mul op_sel_hi:[1,0]",
                           "pk_mul op_sel_hi:[0,0]", "pk_mul op_sel:[0,1] hi:[1,0]",
                           "pk_fma", "pk_fma op_sel:[0,1,0]", "pk_fma op_sel:[1,0,0]", "pk_fma op_sel:[0,0,1]", "pk_fma op_sel_hi:[0,1,1]", "pk_fma op_sel_hi:[1,0,1]",
                           "pk_fma op_sel_hi:[1,1,0]", "pk_fma sel:[0,1,0] hi:[1,0,1]", "pk_fma sel:[1,1,0] hi:[0,0,1]", "pk_fma op_sel:[0,1,0] neg b",
                           "pk_mov op_sel:[0,1]", "pk_mov op_sel:[1,0]", "scalar C++ (ctl)"};
constexpr int NF = 31;

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
  } else if (KIND == 1) {                           // v_mfma_f32_16x16x32_bf16
    v8s as = {1, 2, 3, 4, 5, 6, 7, (short)tid}, bs = {8, 7, 6, 5, 4, 3, 2, 1};
    v8bf a = (v8bf)as, b = (v8bf)bs; v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, b, c3, 0, 0, 0);
      }
    acc = (unsigned)(c0[0] + c1[1] + c2[2] + c3[3]);
  } else if (KIND == 2) {                           // v_mfma_f32_16x16x32_f16
    v8s as = {1, 2, 3, 4, 5, 6, 7, (short)tid}, bs = {8, 7, 6, 5, 4, 3, 2, 1};
    v8h a = (v8h)as, b = (v8h)bs; v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, b, c3, 0, 0, 0);
      }
    acc = (unsigned)(c0[0] + c1[1] + c2[2] + c3[3]);
  } else if (KIND == 3) {                           // v_mfma_f32_32x32x16_bf16
    v8s as = {1, 2, 3, 4, 5, 6, 7, (short)tid}, bs = {8, 7, 6, 5, 4, 3, 2, 1};
    v8bf a = (v8bf)as, b = (v8bf)bs; v16f c0 = {}, c1 = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0); }
    acc = (unsigned)(c0[0] + c1[1]);
  } else if (KIND == 4) {                           // v_mfma_i32_32x32x32_i8
    v4i a = {tid, 1, 2, 3}, b = {4, tid, 6, 7}; v16i c0 = {}, c1 = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u) { c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, c1, 0, 0, 0); }
    acc = c0[0] + c1[1];
  } else if (KIND == 5) {                           // v_mfma_f32_16x16x32_fp8_fp8 (8-byte operands)
    long a = tid * 0x0101010101010101l, b = 0x3838383838383838l; v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b, b, c3, 0, 0, 0);
      }
    acc = (unsigned)(c0[0] + c1[1] + c2[2] + c3[3]);
  } else if (KIND == 6) {                           // v_mfma_f32_16x16x16_bf16 (gfx942 form: 8-byte operands)
    typedef short v4s __attribute__((ext_vector_type(4)));
    v4s a = {1, 2, 3, (short)tid}, b = {4, 3, 2, 1}; v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(b, b, c3, 0, 0, 0);
      }
    acc = (unsigned)(c0[0] + c1[1] + c2[2] + c3[3]);
  } else if (KIND == 7) {                           // v_mfma_f64_16x16x4_f64
    typedef double v4d __attribute__((ext_vector_type(4)));
    double a = tid, b = 1.5; v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
      }
    acc = (unsigned)(c0[0] + c1[1] + c2[2] + c3[3]);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
static const char* CN[] = {"i8 16x16x64", "bf16 16x16x32", "f16 16x16x32", "bf16 32x32x16", "i8 32x32x32", "fp8 16x16x32", "bf16 16x16x16", "f64 16x16x4"};
constexpr int NC = 8;

typedef void (*vk_t)(int, float2*);
typedef void (*ck_t)(int, unsigned*);
template <int... I> static void fill_v(vk_t* t, std::integer_sequence<int, I...>) { ((t[I] = victim<I>), ...); }
template <int... I> static void fill_c(ck_t* t, std::integer_sequence<int, I...>) { ((t[I] = corunner<I>), ...); }

int main(int argc, char** argv) {
  const int blocks = 2048, iters = 8, reps = 3;
  const int vict_lds = argc > 1 ? atoi(argv[1]) : 39488, co_lds = argc > 2 ? atoi(argv[2]) : 81920;
  const size_t words = (size_t)blocks * iters * 512;
  float2* out; unsigned* sink;
  hipMalloc(&out, words * 8); hipMalloc(&sink, 64);
  hipStream_t s1, s2;
  hipStreamCreate(&s1); hipStreamCreate(&s2);
  vk_t vs[NF]; ck_t cs[NC];
  fill_v(vs, std::make_integer_sequence<int, NF>{});
  fill_c(cs, std::make_integer_sequence<int, NC>{});
  std::vector<float2> ref(words), ho(words);
  printf("%-32s", "form \\ co-runner");
  for (int c = 0; c < NC; ++c) printf(" %13s", CN[c]);
  printf("\n");
  for (int v = 0; v < NF; ++v) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(vs[v]), hipFuncAttributeMaxDynamicSharedMemorySize, vict_lds);
    hipLaunchKernelGGL(vs[v], dim3(blocks), dim3(512), vict_lds, s1, iters, out);
    hipStreamSynchronize(s1);
    hipMemcpy(ref.data(), out, words * 8, hipMemcpyDeviceToHost);
    printf("%-32s", FN[v]);
    size_t q[4] = {0, 0, 0, 0}, nlo = 0, nhi = 0;
    char example[512] = "";
    for (int c = 0; c < NC; ++c) {
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
        hipMemcpy(ho.data(), out, words * 8, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < words; ++i) {
          const bool bl = memcmp(&ho[i].x, &ref[i].x, 4) != 0, bh = memcmp(&ho[i].y, &ref[i].y, 4) != 0;
          if (bl || bh) {
            ++bad; ++q[(i & 63) >> 4]; nlo += bl; nhi += bh;
            if (!example[0]) {
              float ax, ay, bx, by;
              const size_t tid = i % 512, blk = i / ((size_t)512 * iters), it = (i / 512) % iters;
              inputs((unsigned)(blk * 512 + tid), (unsigned)it, ax, ay, bx, by);
              snprintf(example, sizeof example, "lane %zu: got (%g, %g) want (%g, %g); a = (%g, %g) b = (%g, %g) [under %s]", i & 63, ho[i].x, ho[i].y, ref[i].x,
                       ref[i].y, ax, ay, bx, by, CN[c]);
            }
          }
        }
      }
      stop = true; ag.join();
      printf(" %13zu", bad);
      fflush(stdout);
    }
    printf("\n");
    if (q[0] + q[1] + q[2] + q[3]) printf("      wrong lanes 0-15: %zu, 16-31: %zu, 32-47: %zu, 48-63: %zu; low half wrong %zu, high half wrong %zu\n      %s\n", q[0], q[1], q[2], q[3], nlo, nhi, example);
  }
  printf("(wrong float2 results of %zu per cell: %d runs of %zu)\n", words * reps, reps, words);
  return 0;
}
