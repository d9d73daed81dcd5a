#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
// C[16][16] += A[16][64] * B[64][16]  (int8 operands), layout probe
__global__ void k(const int8_t* A, const int8_t* B, int* C) {   // A: [16][64] row-major; B given as Bt[16][64] (column j's 64 k-values contiguous)
  const int lane = threadIdx.x;
  const int r = lane & 15, kq = lane >> 4;
  v4i a = *reinterpret_cast<const v4i*>(A + r * 64 + 16 * kq);
  v4i b = *reinterpret_cast<const v4i*>(B + r * 64 + 16 * kq);
  v4i c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  for (int g = 0; g < 4; ++g) C[lane * 4 + g] = c[g];
}
int main() {
  int8_t hA[16 * 64], hB[16 * 64];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 64; ++k) { hA[i * 64 + k] = (int8_t)((i * 7 + k * 3) % 23 - 11); hB[i * 64 + k] = (int8_t)((i * 5 + k * 11) % 19 - 9); }
  int ref[16][16];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int s = 0; for (int k = 0; k < 64; ++k) s += (int)hA[i * 64 + k] * (int)hB[j * 64 + k]; ref[i][j] = s; }
  int8_t *dA, *dB; int* dC; int hC[256];
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 1024);
  hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
  // try the two candidate output layouts
  int okA = 1, okB = 1;
  for (int lane = 0; lane < 64; ++lane) for (int g = 0; g < 4; ++g) {
    int col = lane & 15;
    int rowA = (lane >> 4) * 4 + g;      // f32-style
    int rowB = (lane >> 4) + 4 * g;      // f64-style
    if (hC[lane * 4 + g] != ref[rowA][col]) okA = 0;
    if (hC[lane * 4 + g] != ref[rowB][col]) okB = 0;
  }
  printf("layout f32-style (row = 4*(lane>>4)+g): %d ; f64-style (row = (lane>>4)+4g): %d\n", okA, okB);
  return 0;
}
