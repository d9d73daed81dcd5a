// Synthetic co-runners for the cross-context interference hunt (linked only into tools/hunt/libvipmi_hunt.so).
// Every kernel: 256 threads, dynamic LDS of `lds` bytes, `iters` rounds of one kind of work.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256, 2) void aggressor(int lds_bytes, int iters, const uint4* __restrict__ gbuf, size_t gwords16,
                                                      unsigned* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tid = threadIdx.x;
  const int slots = lds_bytes / 16;                 // 16-byte slots
  uint4* s4 = reinterpret_cast<uint4*>(sm);
  for (int e = tid; e < slots; e += 256) s4[e] = make_uint4(e, tid, blockIdx.x, 7u);
  __syncthreads();
  unsigned acc = 0;
  if (KIND == 0) {                                  // idle: hold the LDS, sleep
    for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(64);
  } else if (KIND == 1) {                           // ds_read_b128 stream
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
      for (int u = 0; u < 8; ++u) {
        const uint4 v = s4[(tid + 256 * (u + 8 * it)) % slots];
        acc += v.x ^ v.w;
      }
    }
  } else if (KIND == 2) {                           // int8 MFMA stream, operands in registers
    v4i a = {tid, 1, 2, 3}, b = {4, tid, 6, 7};
    v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, b, c3, 0, 0, 0);
      }
    }
    acc = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (KIND == 3) {                           // ds_write_b128 stream
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
      for (int u = 0; u < 8; ++u) s4[(tid + 256 * (u + 8 * it)) % slots] = make_uint4(it, u, tid, acc);
    }
  } else if (KIND == 4) {                           // LDS atomics with return
    unsigned* w = reinterpret_cast<unsigned*>(sm);
    const int words = lds_bytes / 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
      for (int u = 0; u < 8; ++u) acc += atomicAdd(&w[(tid * 17 + 4099 * (u + 8 * it)) % words], 1u);
    }
  } else if (KIND == 5) {                           // global loads (L2 / HBM stream)
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
      for (int u = 0; u < 8; ++u) {
        const uint4 v = gbuf[((size_t)blockIdx.x * 2048 + (size_t)(u + 8 * it) * 256 + tid) % gwords16];
        acc += v.x ^ v.z;
      }
    }
  } else if (KIND == 6) {                           // LDS reads + int8 MFMA (the Gram's inner loop without its global side)
    v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const v4i* sv = reinterpret_cast<const v4i*>(sm);
    for (int it = 0; it < iters; ++it) {
      const v4i a = sv[(tid + 256 * (2 * it)) % slots], b = sv[(tid + 256 * (2 * it + 1)) % slots];
      c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, b, c3, 0, 0, 0);
    }
    acc = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (KIND == 7) {                           // barriers + LDS write / read ping-pong
    for (int it = 0; it < iters; ++it) {
      s4[(tid + 256 * it) % slots] = make_uint4(it, tid, 0, acc);
      __syncthreads();
      acc += s4[(255 - tid + 256 * it) % slots].x;
      __syncthreads();
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

extern "C" int vipmi_hunt_aggressor(void* stream, int kind, int lds_bytes, int blocks, int iters, const void* gbuf, size_t gbytes,
                                    void* sink) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const uint4* g = reinterpret_cast<const uint4*>(gbuf);
  unsigned* sk = reinterpret_cast<unsigned*>(sink);
#define AGG(K)                                                                                                             \
  case K:                                                                                                                  \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor<K>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return -2; \
    hipLaunchKernelGGL(aggressor<K>, dim3(blocks), dim3(256), lds_bytes, st, lds_bytes, iters, g, gbytes / 16, sk);       \
    break;
  switch (kind) {
    AGG(0) AGG(1) AGG(2) AGG(3) AGG(4) AGG(5) AGG(6) AGG(7)
    default: return -1;
  }
#undef AGG
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
