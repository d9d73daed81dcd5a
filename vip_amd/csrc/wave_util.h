// wave_util.h -- full-wave (64 lanes) float64 reductions on DPP, shared by the eigensolvers.
#pragma once
#include <hip/hip_runtime.h>

namespace vipmi {

// Full-wave (64 lanes) sum of a double, result in every lane, without touching the LDS crossbar:
// xor-butterflies inside a 16-lane row with DPP (quad_perm xor1/xor2, row_half_mirror, row_mirror --
// all lanes of a row then hold the row sum), then the four row sums are read with v_readlane and added.
// (__shfl_xor on a double is two ds_bpermute round trips per step: ~18 dependent LDS round trips for
// the three dot products of one rotation, which dominated the eigensolver.)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]  (xor 1)
  v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]  (xor 2)
  v += dpp_f64<0x141>(v);   // row_half_mirror      (i <-> 7-i)
  v += dpp_f64<0x140>(v);   // row_mirror           (i <-> 15-i)
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// Eight full-wave sums at once, "transposed": lane l returns the sum over the 64 lanes of x[l & 7].  Each of the first three
// stages halves the number of values a lane carries (the lane keeps the values whose index bit equals its lane bit and sends
// the others to its partner: quad_perm xor 1, xor 2, then a rotation by 4 inside the row of 16 -- the sender of a rotation has
// the opposite bit 2, which is all the scheme needs); the last three stages (rotation by 8, v_permlane16_swap,
// v_permlane32_swap -- gfx950) run on ONE value.  ~67 instructions against 8 x 23 for eight wave_sum()s, one dependent chain of
// six exchanges instead of eight; no SGPR traffic.  The summation order differs from wave_sum (not bit-identical to it).
__device__ __forceinline__ double swap_add16_f64(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double swap_add32_f64(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double wave_sum8_scatter(const double (&x)[8]) {
  const int lane = threadIdx.x & 63;
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
  double y[4], z[2];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const double keep = b0 ? x[2 * m + 1] : x[2 * m], send = b0 ? x[2 * m] : x[2 * m + 1];
    y[m] = keep + dpp_f64<0xB1>(send);
  }
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const double keep = b1 ? y[2 * m + 1] : y[2 * m], send = b1 ? y[2 * m] : y[2 * m + 1];
    z[m] = keep + dpp_f64<0x4E>(send);
  }
  const double keep = b2 ? z[1] : z[0], send = b2 ? z[0] : z[1];
  double w = keep + dpp_f64<0x124>(send);     // row_ror:4
  w += dpp_f64<0x128>(w);                     // row_ror:8
  w = swap_add16_f64(w);
  return swap_add32_f64(w);
}

__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_f64<0xB1>(v));
  v = fmax(v, dpp_f64<0x4E>(v));
  v = fmax(v, dpp_f64<0x141>(v));
  v = fmax(v, dpp_f64<0x140>(v));
  return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// Columns cross workgroups only through global memory.  They are written with agent-scope write-through
// stores (global_store_dwordx2 sc1) and read back with sc1 loads (L1 bypassed), so the barrier needs no
// release/acquire fences (no buffer_wbl2 / buffer_inv): every storing wave drains its stores, one lane
// bumps the counter and polls it (MI355X guide, G16 recipe R1).
__device__ __forceinline__ double ld_shared(const double* p) {
  unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
  return __longlong_as_double((long long)u);
}
__device__ __forceinline__ void st_shared(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Bounded spinning.  A partner that never became resident must not hang the GPU: after 2^22 polls (seconds) the wait is given up,
// which is latched in the context's deferred-failure words (fail[1]: vipmi_check_deferred) -- and in fail[2], "this launch is
// dead" (cleared by deferred_fail_words() before every launch): every later barrier of the launch looks at that word once per 1024
// polls and gives up at once, so a dead launch drains in milliseconds instead of timing out again at each of its ~n steps
// (round 4: one such launch held the GPU for the rest of a test run).
__device__ __forceinline__ bool barrier_gave_up(unsigned spins, int* report, int* fail = nullptr) {
  int* f = fail ? fail : report;
  if (!f) return spins > (1u << 22);
  if ((spins & 1023u) == 0u && __hip_atomic_load(f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
  if (spins > (1u << 22)) {
    if (report) {
      atomicAdd(report + 1, 1);
      atomicAdd(report + 2, 1);          // non-zero: the launch is dead; the count is what a recovery launch takes back from [1]
    }
    return true;
  }
  return false;
}

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, int nwg, int* fail = nullptr) {
  if (nwg == 1) {
    __syncthreads();
    return;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its write-through stores have landed
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (barrier_gave_up(++spins, fail)) break;
    }
  }
  __syncthreads();
}

// ---- the same exchange between workgroups of ONE XCD ------------------------------------------------------------------------
// An agent-scope (sc1) store leaves the XCD's L2 for memory, and every reader pays a fabric round trip (~1 us) whichever XCD
// it sits on.  Workgroups that share an XCD share its L2: a workgroup-scope store (sc0: written through the CU's L1, KEPT in
// the L2) is visible to an sc1 load (L1 bypassed, L2-served) of any CU of that XCD at L2 latency.  There is no "XCD" scope in
// the memory model, so the kernel that uses this (a) places its workgroups itself -- workgroup ids go round-robin over the 8
// XCDs --, (b) VERIFIES the placement at run time with HW_REG_XCC_ID behind one agent-scope barrier, and (c) falls back to the
// agent-scope exchange when the check fails.  The barrier is a flag per workgroup (plain epoch numbers, no atomic: an atomic
// is executed beyond the L2), polled by one lane per flag.
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(6164) & 15u; }   // hwreg(HW_REG_XCC_ID, 0, 4)
__device__ __forceinline__ void st_xcd(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void xcd_barrier(unsigned* flags, unsigned epoch, int nwg, int wg, int* fail = nullptr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its stores are in the L2
  __syncthreads();
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(flags + wg, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    const int l = threadIdx.x;
    unsigned spins = 0;
    bool ok = l >= nwg;
    while (true) {
      if (!ok) ok = (int)(__hip_atomic_load(flags + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) >= 0;
      if (__all(ok)) break;
      if (__any(barrier_gave_up(++spins, threadIdx.x == 0 ? fail : nullptr, fail))) break;
    }
  }
  __syncthreads();
}

}  // namespace vipmi
