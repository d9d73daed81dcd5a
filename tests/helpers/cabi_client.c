/* A plain C client of the C ABI (include/vipmi.h): no Python, no PyTorch -- HIP runtime calls for the device buffers
 * only.  Reads a float32 cube and float64 angles from a raw file written by the test, runs vipmi_pca_fullframe_f32 and
 * writes the final frame back.  Built and driven by tests/test_gpu_pca.py::test_c_client_of_the_c_abi.
 *   usage: cabi_client <in.bin> <out.bin>      in.bin = int64 n, int64 N, int64 ncomp, float32 cube[n*N*N], float64 angles[n] */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "vipmi.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_VIPMI(x) do { int s_ = (x); if (s_ != VIPMI_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, s_, vipmi_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
  if (argc != 3) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  int64_t hdr[3];
  if (fread(hdr, sizeof(int64_t), 3, f) != 3) return 1;
  const int64_t n = hdr[0], N = hdr[1], k = hdr[2];
  const size_t ncube = (size_t)n * N * N;
  float* cube = (float*)malloc(ncube * sizeof(float));
  double* angles = (double*)malloc((size_t)n * sizeof(double));
  float* frame = (float*)malloc((size_t)N * N * sizeof(float));
  if (fread(cube, sizeof(float), ncube, f) != ncube || fread(angles, sizeof(double), (size_t)n, f) != (size_t)n) return 1;
  fclose(f);

  float *d_cube = NULL, *d_frame = NULL;
  CHECK_HIP(hipSetDevice(0));
  CHECK_HIP(hipMalloc((void**)&d_cube, ncube * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_frame, (size_t)N * N * sizeof(float)));
  CHECK_HIP(hipMemcpy(d_cube, cube, ncube * sizeof(float), hipMemcpyHostToDevice));

  vipmi_ctx* ctx = NULL;
  CHECK_VIPMI(vipmi_create(0, NULL, &ctx));
  CHECK_VIPMI(vipmi_pca_fullframe_f32(ctx, d_cube, angles, n, N, k, 0, NULL, VIPMI_COLLAPSE_MEDIAN, d_frame, NULL, NULL, NULL,
                                      NULL));
  CHECK_VIPMI(vipmi_synchronize(ctx));
  CHECK_HIP(hipMemcpy(frame, d_frame, (size_t)N * N * sizeof(float), hipMemcpyDeviceToHost));
  CHECK_VIPMI(vipmi_destroy(ctx));
  CHECK_HIP(hipFree(d_cube));
  CHECK_HIP(hipFree(d_frame));

  f = fopen(argv[2], "wb");
  if (!f) return 1;
  fwrite(frame, sizeof(float), (size_t)N * N, f);
  fclose(f);
  printf("vipmi %d: frame of %ld x %ld written\n", vipmi_version(), (long)N, (long)N);
  return 0;
}
