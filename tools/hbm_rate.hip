// HBM streaming rates on the box: read-only (sum), copy, for a 419 MB / 1.68 GB buffer.   hipcc --offload-arch=gfx950 -O3 tools/hbm_rate.hip -o /tmp/hbm_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL>
__global__ __launch_bounds__(256) void rd(const f32x4* __restrict__ a, int64_t n4, float* out) {
  f32x4 s = {0, 0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * 256 * UNROLL;
  for (int64_t i = (int64_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n4; i += stride) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = (i + 256 * u < n4) ? a[i + 256 * u] : f32x4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) s += v[u];
  }
  if (s.x + s.y + s.z + s.w == 1.2345f) out[0] = 1.f;
}
template <int UNROLL>
__global__ __launch_bounds__(256) void cp(const f32x4* __restrict__ a, f32x4* __restrict__ b, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256 * UNROLL;
  for (int64_t i = (int64_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n4; i += stride) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = (i + 256 * u < n4) ? a[i + 256 * u] : f32x4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) if (i + 256 * u < n4) b[i + 256 * u] = v[u];
  }
}
int main() {
  for (int64_t bytes : {(int64_t)419430400, (int64_t)1677721600}) {
    f32x4 *a, *b; float* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    const int64_t n4 = bytes / 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1024, 2048, 4096, 16384}) {
      float best_r = 1e9, best_c = 1e9, ms;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(rd<8>, dim3(grid), dim3(256), 0, 0, a, n4, o); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); if (ms < best_r) best_r = ms;
        hipEventRecord(e0); hipLaunchKernelGGL(cp<8>, dim3(grid), dim3(256), 0, 0, a, b, n4); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); if (ms < best_c) best_c = ms;
      }
      printf("%5.0f MB grid %5d: read %.3f ms = %.2f TB/s   copy %.3f ms = %.2f TB/s (read + write)\n", bytes / 1e6, grid, best_r, bytes / best_r / 1e9, best_c,
             2.0 * bytes / best_c / 1e9);
    }
    hipFree(a); hipFree(b); hipFree(o);
  }
  return 0;
}
