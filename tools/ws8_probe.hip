// wave_sum8_scatter (wave_util.h) against host sums.  build: hipcc --offload-arch=gfx950 -O3 tools/ws8_probe.hip -o gpurun_out/ws8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../vip_amd/csrc/wave_util.h"
__global__ void k(const double* in, double* out) {
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = in[i * 64 + threadIdx.x];
  out[threadIdx.x] = vipmi::wave_sum8_scatter(x);
  out[64 + threadIdx.x] = vipmi::wave_sum(x[threadIdx.x & 7]);
}
int main() {
  std::vector<double> h(512), o(128);
  for (int i = 0; i < 512; ++i) h[i] = sin(0.37 * i) * (1 + i % 7);
  double *din, *dout;
  hipMalloc(&din, 512 * 8); hipMalloc(&dout, 128 * 8);
  hipMemcpy(din, h.data(), 512 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
  hipMemcpy(o.data(), dout, 128 * 8, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int l = 0; l < 64; ++l) {
    double ref = 0;
    for (int j = 0; j < 64; ++j) ref += h[(l & 7) * 64 + j];
    worst = fmax(worst, fabs(o[l] - ref));
  }
  printf("wave_sum8_scatter max |err| = %.3e (%s)\n", worst, worst < 1e-12 ? "OK" : "WRONG");
  return worst < 1e-12 ? 0 : 1;
}
