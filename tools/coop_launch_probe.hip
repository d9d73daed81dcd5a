// Launch cost of hipLaunchCooperativeKernel against a plain launch for the shape of tri_wave_kernel (64 single-wave workgroups):
// VERDICT r4 #4 asked for the evaluation.   hipcc --offload-arch=gfx950 -O3 tools/coop_launch_probe.hip -o /tmp/coop && /tmp/coop
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k_plain(unsigned* p) { if (threadIdx.x == 0) atomicAdd(p, 1u); }
__global__ __launch_bounds__(64) void k_coop(unsigned* p) {
  if (threadIdx.x == 0) atomicAdd(p, 1u);
  cooperative_groups::this_grid().sync();
  if (threadIdx.x == 0) atomicAdd(p + 1, 1u);
}
int main() {
  unsigned* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int coop = 0; hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, 0);
  printf("cooperative launch supported: %d\n", coop);
  const int N = 2000;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, st);
      for (int i = 0; i < N; ++i) {
        if (mode == 0) hipLaunchKernelGGL(k_plain, dim3(64), dim3(64), 0, st, d);
        else { void* args[] = {&d}; hipError_t e = hipLaunchCooperativeKernel((const void*)k_coop, dim3(64), dim3(64), args, 0, st); if (e != hipSuccess) { printf("coop launch failed: %s\n", hipGetErrorString(e)); return 1; } }
      }
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%s: %.2f us per back-to-back launch (64 x 64 threads)\n", mode ? "hipLaunchCooperativeKernel + grid.sync" : "plain launch", ms * 1e3 / N);
    }
  }
  return 0;
}
