// Issue rate / dependent latency of the packed-FP32 forms the FFT shears are made of, at 1 / 2 / 3 waves per SIMD, on the whole chip
// (the clock under a chip-filling packed stream is part of the answer).   hipcc --offload-arch=gfx950 -O3 tools/pk_issue_probe.hip -o /tmp/pkp
//   dep     : one chain of dependent v_pk_fma_f32
//   cmul    : chains of (v_pk_mul_f32 ; v_pk_fma_f32 on its result) -- the complex multiply of fft_wave.h -- with ILP independent chains
//   add     : ILP independent v_pk_add_f32
//   lds     : cmul stream with a ds_write2_b64 / ds_read2_b64 pair every 16 packed instructions (the exchange density of the shears)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float cf __attribute__((ext_vector_type(2)));
constexpr int ITER = 2000;

template <int MODE, int ILP>
__global__ __launch_bounds__(1024) void probe(cf* out, unsigned long long* cyc, cf seed) {
  extern __shared__ cf lds[];
  cf a[8], w = seed;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = cf{seed.x + i, seed.y - i} * (1.f + threadIdx.x * 1e-6f);
  cf* slot = lds + threadIdx.x;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITER; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[0]) : "v"(w));
    } else if (MODE == 1 || MODE == 3) {
#pragma unroll
      for (int r = 0; r < 16 / ILP; ++r)
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
          cf d;
          asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
                       "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
                       : "=&v"(d) : "v"(a[i]), "v"(w));
          a[i] = d;
        }
      if (MODE == 3) {
        slot[0] = a[0]; slot[1100] = a[1];
        __builtin_amdgcn_wave_barrier();
        a[2] = a[2] + slot[64 ^ 1]; a[3] = a[3] + slot[1100 + (64 ^ 1)];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 32 / ILP; ++r)
#pragma unroll
        for (int i = 0; i < ILP; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  cf s = a[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int ILP>
void run(const char* name, int waves_per_simd, int grid) {
  cf* out; unsigned long long* cyc;
  const int threads = 256 * waves_per_simd;
  hipMalloc(&out, sizeof(cf) * grid * threads); hipMalloc(&cyc, 8 * grid);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, ILP>), dim3(grid), dim3(threads), 32768, 0, out, cyc, cf{1.0001f, 1e-4f});
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double pk_per_wave = (MODE == 2 ? 32.0 : 16.0 * (MODE == 0 ? 1 : 2)) * ITER;
  // SIMD-time per packed instruction: time x clock / (instructions per SIMD); report ns so no clock assumption is needed
  const double ns_per_pk_simd = best * 1e6 / (pk_per_wave * waves_per_simd);
  printf("%-10s ILP %d  waves/SIMD %d grid %4d: %.3f ms  -> %.3f ns per packed instruction and SIMD (4 cycles at 2.4 GHz = 1.67 ns, at 1.85 GHz = 2.16 ns)\n",
         name, ILP, waves_per_simd, grid, best, ns_per_pk_simd);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int grid : {256, 64}) {
    for (int w : {1, 2, 3, 4}) {
      run<0, 1>("dep", w, grid);
      run<1, 1>("cmul", w, grid);
      run<1, 2>("cmul", w, grid);
      run<1, 4>("cmul", w, grid);
      run<1, 8>("cmul", w, grid);
      run<2, 8>("add", w, grid);
      run<3, 8>("cmul+lds", w, grid);
    }
  }
  return 0;
}
