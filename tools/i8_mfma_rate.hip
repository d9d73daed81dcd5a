// cycles per v_mfma_i32_16x16x64_i8 (and 32x32x32) issued back to back on one SIMD, 1 / 2 / 4 waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ void k(int* out, int iters) {
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
  v4i c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  v16i d0 = {}, d1 = {}, d2 = {}, d3 = {};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c5, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c7, 0, 0, 0);
    } else {
      d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d3, 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  int s = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3] + d0[0] + d1[1] + d2[2] + d3[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (int)(t1 - t0); out[1] = s; }
}
int main() {
  int* d; hipMalloc(&d, 64); int h[2];
  for (int kind = 0; kind < 2; ++kind)
    for (int waves = 4; waves <= 16; waves *= 2) {          // waves per workgroup = per CU -> waves / 4 per SIMD
      const int iters = 20000;
      if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 0, 0, d, iters); else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, d, iters);
      hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
      const int per_iter = kind == 0 ? 8 : 4;
      printf("%s, %d wave(s) per SIMD: %.1f s_memtime ticks per MFMA per wave (%.1f per SIMD-MFMA)\n", kind == 0 ? "16x16x64" : "32x32x32", waves / 4,
             (double)h[0] / iters / per_iter, (double)h[0] / iters / per_iter / (waves / 4));
    }
  // whole chip, wall clock: W workgroups of 256 threads (one wave per SIMD each); 256 -> one per CU, 1024 -> four per CU
  for (int wgs = 256; wgs <= 2048; wgs *= 2) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)wgs * 4 * iters * 8 * 32768.0;
    printf("16x16x64 i8, %d workgroups x 4 waves: %.3f ms -> %.0f TOPS\n", wgs, ms, ops / ms / 1e9);
  }
  return 0;
}
