// common.h -- internal helpers shared by the libvipmi translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <chrono>
#include <vector>
#include "../../include/vipmi.h"

// ---- gfx950: a packed-FP32 operand form that must not be emitted (measured, tools/hunt/probe4.hip / probe5.hip) -----------------
// v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 whose LOW result takes src0 from its low half and src1 from its HIGH half
// (`op_sel:[0,1]`, `op_sel:[0,1,x]`, any op_sel_hi / neg) return a wrong low half in lanes 48..63 -- src1 reads as 0 -- while
// ANOTHER wave of the CU executes v_mfma_i32_16x16x64_i8 / v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x32_f16 (the 16-byte-operand
// 16x16 MFMAs new on gfx950; the 32x32 forms, the 8-byte-operand forms, f32 and f64 MFMA do not trigger it).  Every other
// selection is exact, including the mirrored one (`op_sel:[1,0]`): the operation is commutative, so the swizzle always moves to
// src0.  This is what made the median wrong beside another context's int8 Gram product (round 3's open bug): hipcc had emitted
// `v_pk_add_f32 v[a:b], v[a:b], v[c:d] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]` for the bin index (x - lo) * scale.
//  * hand-written packed math (fft_wave.h, derotate_direct2.hip) puts the swizzled operand in src0;
//  * kernels with compiler-generated float math carry VIPMI_NO_PK32 (no packed-FP32 selection at all) when the compiler is
//    seen to emit the form -- tools/isa_lint.py scans the device code of every object and of libvipmi.so (Makefile `lint`
//    target, tests/test_isa_lint.py) and fails on any `op_sel:[0,1` of a v_pk_*_f32.
#if defined(__HIP_DEVICE_COMPILE__)
#define VIPMI_NO_PK32 __attribute__((target("no-packed-fp32-ops")))
#else
#define VIPMI_NO_PK32           /* (the host pass does not know the AMDGPU feature) */
#endif

namespace vipmi {

void set_error(const char* fmt, ...);

#define VIPMI_CHECK_HIP(expr)                                                          \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      vipmi::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return VIPMI_ERR_HIP;                                                            \
    }                                                                                  \
  } while (0)

#define VIPMI_REQUIRE(cond, ...)                  \
  do {                                            \
    if (!(cond)) {                                \
      vipmi::set_error(__VA_ARGS__);              \
      return VIPMI_ERR_ARG;                       \
    }                                             \
  } while (0)

#define VIPMI_TRY(expr)            \
  do {                             \
    int _s = (expr);               \
    if (_s != VIPMI_OK) return _s; \
  } while (0)

// Named scratch buffers that live as long as the ctx (grow-only).
struct Buffer {
  void* ptr = nullptr;
  size_t bytes = 0;
};

// accumulating stage timer: every tic/toc pair since the last reset is kept (hipEvents on the ctx stream)
struct StageTimer {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  int used = 0;
  bool open = false;
  double host_ms = 0.0;            // option timing = 3: host wall time spent enqueuing the stage (no events)
  int host_count = 0;
  std::chrono::steady_clock::time_point host_t0;
};

}  // namespace vipmi

struct vipmi_gate {
  std::mutex mu;
  std::vector<hipEvent_t> ring;     // completion events of the gated sections, reused round-robin
  int next = 0;
  hipEvent_t last = nullptr;
};

struct vipmi_ctx {
  vipmi_gate* gate = nullptr;
  bool gate_armed = false;          // set by the fused pca call: its project stage waits for the previous section
  int device = 0;
  hipStream_t stream = nullptr;
  std::map<std::string, vipmi::Buffer> buffers;
  std::map<std::string, vipmi::StageTimer> timers;
  std::map<std::string, int64_t> options;
  std::map<std::string, std::string> upload_keys;
  struct PinnedSlot {
    void* host = nullptr;
    size_t bytes = 0;
    hipEvent_t ev = nullptr;
  };
  std::map<std::string, std::vector<PinnedSlot>> pinned;   // small rings of pinned staging buffers
  std::map<std::string, int> pinned_next;
  void* host_pinned = nullptr;     // grow-only pinned scratch for small read-backs (host_scratch)
  size_t host_pinned_bytes = 0;
  hipStream_t copy_stream = nullptr;       // host -> device copies that run beside the kernels of `stream` (vipmi_pca_fullframe_hostin_f32)
  std::vector<hipEvent_t> copy_events;
  const void* gram_given_ref = nullptr;    // set by the host-input front: workspace "pca_G" already holds ref ref^T for this matrix
  int64_t gram_given_n = 0;
  int sticky_fail[2] = {0, 0};    // deferred failures latched on the host when their device words are freed (vipmi_trim)
  int num_cu = 256;
  int timing = 0;                 // 0 off, 1 every stage / kernel, 2 only the roofline kernel (k_rot_s2)

  // returns a device buffer of at least `bytes` (contents undefined)
  int get(const char* name, size_t bytes, void** out);
  // device copy of a small host table, re-uploaded only when `key` changes (synchronous upload)
  // true (and *out set) when workspace `name` already holds the upload tagged `key` -- lets callers skip building the
  // host table at all (the twiddle tables cost ~0.1 ms of host time per call otherwise)
  bool cached(const char* name, const std::string& key, void** out) {
    auto it = upload_keys.find(name);
    if (it == upload_keys.end() || it->second != key) return false;
    auto b = buffers.find(name);
    if (b == buffers.end() || !b->second.ptr) return false;
    *out = b->second.ptr;
    return true;
  }
  int upload_cached(const char* name, const std::string& key, const void* host, size_t bytes,
                    void** out);
  // asynchronous H2D of a small host table through a ring of pinned staging buffers (never blocks the
  // host unless the ring wraps onto a copy that is still in flight)
  int upload_async(const char* name, const void* host, size_t bytes, void* dst);
  // pinned host memory of at least `bytes` (contents undefined; valid until the next call that asks for more)
  int host_scratch(size_t bytes, void** out);
  int64_t opt(const char* key, int64_t dflt) const {
    auto it = options.find(key);
    return it == options.end() ? dflt : it->second;
  }
  // gated section (see vipmi_set_gate): enter = wait for the previous section, leave = publish this one's end
  int gate_enter();
  int gate_leave();
  void tic(const char* stage);
  void toc(const char* stage);
};

namespace vipmi {

// incremental int8 Gram matrix (gram_i8.hip)
struct GramI8Inc {
  int64_t n = 0, P = 0, ld = 0, klen = 0, Ppad = 0, plane = 0;
  int npad = 0, nt = 0, nslices = 0, nwg = 0;
  int8_t* D = nullptr;
  double* sc = nullptr;
  double* partial = nullptr;
  int2* d_tiles = nullptr;
};
int gram_i8_inc_begin(vipmi_ctx* ctx, int64_t n, int64_t P, int64_t ld, GramI8Inc* st);
int gram_i8_inc_block(vipmi_ctx* ctx, const GramI8Inc& st, const float* M, int block);
int gram_i8_inc_end(vipmi_ctx* ctx, const GramI8Inc& st, double* G);
bool gram_i8_default_path(vipmi_ctx* ctx, int64_t n, int64_t P);

template <typename T>
inline int ws(vipmi_ctx* ctx, const char* name, size_t count, T** out) {
  void* p = nullptr;
  int s = ctx->get(name, count * sizeof(T), &p);
  *out = reinterpret_cast<T*>(p);
  return s;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// The context's deferred-failure words on the device: [0] eigenproblems that did not converge, [1] inter-workgroup barriers that
// timed out ([2]: scratch of the barriers themselves, wave_util.h barrier_gave_up) (a co-resident partner never arrived: results of that launch are invalid).  Read and cleared by vipmi_check_deferred;
// vipmi_trim latches them on the host before it frees the buffer.
inline int deferred_fail_words(vipmi_ctx* ctx, int** out, bool clear_gave_up = true) {
  const bool fresh = ctx->buffers.find("deferred_fail") == ctx->buffers.end();
  VIPMI_TRY(ws(ctx, "deferred_fail", 4, out));
  if (fresh) VIPMI_CHECK_HIP(hipMemsetAsync(*out, 0, 4 * sizeof(int), ctx->stream));
  else if (clear_gave_up) VIPMI_CHECK_HIP(hipMemsetAsync(*out + 2, 0, sizeof(int), ctx->stream));    // [2]: "the current launch gave up a barrier"
  return VIPMI_OK;
}

// hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), remembered per (device, kernel): the attribute only
// ever has to grow, and the call costs ~3 us -- eight of them were a tenth of the host time of a pca() call on a small cube.
inline hipError_t set_dyn_lds(const void* f, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> seen;
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = seen.find({dev, f});
    if (it != seen.end() && it->second >= bytes) return hipSuccess;
  }
  const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lk(mu);
    int& v = seen[{dev, f}];
    if (v < bytes) v = bytes;
  }
  return e;
}

struct StageScope {
  vipmi_ctx* c;
  const char* s;
  StageScope(vipmi_ctx* ctx, const char* stage) : c(ctx), s(stage) { c->tic(s); }
  ~StageScope() { c->toc(s); }
};

// ---- internal entry points implemented per .hip file (all enqueue on ctx->stream) ----
int gram_f32(vipmi_ctx* ctx, const float* A, int64_t na, const float* B, int64_t nb, int64_t P,
             int64_t ld, double* G);
// symmetric Gram matrices on the int8 matrix cores (gram_i8.hip): mode 1 = 5 digits (7e-12), 2 = 6 digits (2e-15)
int gram_i8_f32(vipmi_ctx* ctx, const float* M, int64_t n, int64_t P, int64_t ld, double* G, int64_t batch, int mode);
// the Gram matrices of the column segments of one matrix (every segment a whole number of K-slices of klen columns): gram_i8.hip
int gram_i8_ragged_f32(vipmi_ctx* ctx, const float* M, int64_t n, int64_t Ptot, int64_t klen, const int32_t* seg_slice,
                       int64_t nseg, double* G_all);
// R = W M for the segments of M side by side, written THROUGH a pixel list into a cube (project.hip): Wt_all [nseg][n][kld]
int rowspace_scatter_f32(vipmi_ctx* ctx, const float* Wt_all, int kld, const float* M, int64_t n, int64_t Ptot,
                         const int32_t* tile_seg, const int32_t* pix_out, const int* frange_all, int64_t P, float* out,
                         const float* rho_all = nullptr, const float* mu32 = nullptr);
int eigh_f64(vipmi_ctx* ctx, double* G, int64_t batch, int64_t n, double* evals, double* evecs);
// leading k eigenpairs only (eigh_tri.hip); falls back to eigh_f64 when the sizes are outside its range or
// option "eigh_method" == 1.  nact: optional device array with the active size of each (zero padded) problem.
bool eigh_topk_supported(int64_t n, int64_t k);
int eigh_topk_f64(vipmi_ctx* ctx, double* G, int64_t batch, int64_t n, int64_t k, const int32_t* nact,
                  double* evals, double* evecs, bool all_evals = false);
int rotate_interp_f32(vipmi_ctx* ctx, const float* in, const double* angles_host, int64_t n, int64_t N, double cx,
                      double cy, int interp, int border, float* out);
int lincomb_f32(vipmi_ctx* ctx, const float* x, const float* y, float a, float b, int64_t total, float* out);
// batched C[b] = A0[ia[b]] B0[ib[b]]^T - A1[ia[b]] B1[ib[b]]^T (bgemm.hip)
int bgemm_abt_f32(vipmi_ctx* ctx, const float* A0, const float* B0, const float* A1, const float* B1,
                  const int32_t* ia, const int32_t* ib, int64_t nbatch, int64_t M, int64_t N, int64_t K, int64_t lda,
                  int64_t ldb, int64_t ldc, int64_t sa, int64_t sb, int64_t sc, float* C);
// Launches whose workgroups WAIT FOR EACH OTHER inside one XCD (the one-XCD layout of tri_multi_kernel: up to 32 whole-CU
// workgroups = the whole XCD; tri_wave_kernel: 64 spinning waves) must not overlap with another such launch on the same XCD: two of
// them, each resident in part, wait for workgroups that no longer fit -- until the barrier time-out.  Several host threads on
// their own streams can produce exactly that (the XCD rotates per launch, but two of four concurrent launches share one more
// often than not).  CoopOrder chains such launches per device: the constructor makes the context's stream wait for the event the
// previous one recorded, done() records this one's.  A lone caller never waits; the lock is held only while enqueuing.
struct CoopOrder {
  vipmi_ctx* ctx;
  int status;
  bool locked;
  explicit CoopOrder(vipmi_ctx* c);
  int done();
  ~CoopOrder();
};
// Householder tridiagonalisation of one matrix of 129 .. 448 rows on 64 cooperating single-wave workgroups, matrix in registers
// (eigh_wave.hip): d, e, tau -> det[3][n], reflectors in the rows of A.  bars: 136 zeroed words; gbuf: 4 * 64 * ceil(n / 64) + 8
// doubles; xcd_slot = 1 + XCD the waves sit on (0 = spread over the chip, agent-scope exchange); fail: deferred-failure words.
bool tri_wave_supported(int64_t n);
bool tri_wave_fits(vipmi_ctx* ctx, int64_t n);   // its 64 single-wave workgroups can be co-resident on ONE XCD of this device (occupancy query, cached)
int tri_wave_reduce(vipmi_ctx* ctx, double* A, int n, double* det, double* gbuf, unsigned* bars, int xcd_slot, int* fail,
                    double* gram,       // gram[ceil((n-2)/4)][8]: products of the reflectors inside each group of four
                    double* det2);      // det2[3 n + 3]: d, e, e^2 scaled to max-norm 1, then scale and the Gershgorin interval
// stages 2-5 from those arrays: eigenvalue (multisection), inverse iteration and back-transformation with one workgroup per
// vector, then Gram-Schmidt + sign convention; leading k <= 64 pairs -> evals[k], evecs[k][n]
int tri_wave_vectors(vipmi_ctx* ctx, const double* A, int n, int k, const double* det, const double* det2, const double* gram,
                     double* evals, double* evecs, const unsigned* bars);   // bars[4] != 0 (tri_wave_reduce timed out): NaN results
// one larger problem (512 < n <= 2048): eigh_tri_large.hip
bool eigh_large_supported(int64_t n, int64_t k);
int eigh_large_f64(vipmi_ctx* ctx, double* G, int64_t batch, int64_t n, int64_t k, double* evals, double* evecs,
                   bool all_evals = false);
int center_f64(vipmi_ctx* ctx, const double* M, int64_t n, int64_t P, int mode, float* D, double* mu, float* mu32);
int gram_offset_f64(vipmi_ctx* ctx, const float* D, const double* mu, int64_t n, int64_t P, double* G, const double* u = nullptr);
int spat_center_f64(vipmi_ctx* ctx, const double* M, int64_t n, int64_t P, int64_t Preal, int with_std, float* D, double* mu, float* mu32,
                    double* u);
int pca_fullframe_f64(vipmi_ctx* ctx, const double* cube, const double* angles_host, int64_t n, int64_t N, int64_t ncomp, int scaling,
                      const uint8_t* mask, int collapse_mode, float* frame, float* pcs, float* recon, float* residuals,
                      float* residuals_der);
int annular_eigh_f64(vipmi_ctx* ctx, const double* G, int64_t nseg, int64_t n, const int32_t* lib_idx, const int32_t* lib_len,
                     int64_t m, int64_t k, double* work, double* evals, double* evecs);
int annular_subgrams_f64(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* lib_idx,
                         const int32_t* lib_len, int64_t max_lib, int64_t m, double* G, double* H);
int annular_apply_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* lib_idx,
                      const int32_t* lib_len, int64_t max_lib, int64_t m, const double* G, const double* evals,
                      const double* evecs, const int32_t* ncomps, int64_t nk, float* residuals, const float* mu32 = nullptr,
                      const double* u = nullptr);
int annular_gram_all_f32(vipmi_ctx* ctx, const float* cube, int64_t n, int64_t P, const int32_t* pix_all, int64_t Ptot, int64_t klen,
                         const int32_t* seg_slice, int64_t nseg, float* A_all, double* G_all);
int annular_apply_all_f32(vipmi_ctx* ctx, const float* A_all, int64_t n, int64_t Ptot, const int32_t* tile_seg, const int32_t* pix_out,
                          int64_t nseg, const int32_t* lib_idx, const int32_t* lib_len, int64_t m, const double* G_all,
                          const double* evals, const double* evecs, const int32_t* kseg, int64_t kmax, int64_t P, float* cube_out,
                          const float* mu32 = nullptr);
int annular_gram_all_f64(vipmi_ctx* ctx, const double* cube, int64_t n, int64_t P, const int32_t* pix_all, int64_t Ptot, int64_t klen,
                         const int32_t* seg_slice, int64_t nseg, int mode, float* D_all, double* mu, float* mu32, double* G_all);
bool eigh_gather_supported(int64_t m, int64_t k);
int eigh_topk_gather_f64(vipmi_ctx* ctx, const double* G, int64_t nseg, int64_t per_seg, int64_t ldg, const int32_t* idx,
                         const int32_t* len, int64_t m, int64_t k, double* work, double* evals, double* evecs);
int eigh_leading(vipmi_ctx* ctx, double* G, int64_t batch, int64_t n, int64_t k, const int32_t* nact,
                 double* evals, double* evecs, bool all_evals = false);
// verified fast path for the leading pairs of ONE positive semi-definite matrix (eigh_chfsi.hip); G is not modified
bool eigh_chfsi_supported(int64_t n, int64_t k);
int64_t eigh_chfsi_pays_from(int64_t k);      // smallest n at which the fast path beats the exact solvers (measured, by block width)
int eigh_chfsi_f64(vipmi_ctx* ctx, const double* G, int64_t n, int64_t k, double* evals, double* evecs, int* converged,
                   int* info);
int rowspace_gemm_f32(vipmi_ctx* ctx, const float* W, const float* M, int64_t k, int64_t n, int64_t P,
                      const float* rowscale, float* B, const int* frange = nullptr);
int subtract_gemm_f32(vipmi_ctx* ctx, const float* M, const float* C, const float* B, int64_t n,
                      int64_t k, int64_t P, float* R, float* recon);
// the same two with the small operand already in the layout the kernels read (project.hip): Wt [n][kld], Ct [k][nld]
int rowspace_gemm_t(vipmi_ctx* ctx, const float* Wt, int kld, const float* M, int64_t k, int64_t n, int64_t P, const float* rowscale,
                    float* T, const int* frange = nullptr);
int subtract_gemm_t(vipmi_ctx* ctx, const float* M, const float* Ct, int nld, const float* T, int64_t n, int64_t k, int64_t P, float* R,
                    float* recon);
int scale_f32(vipmi_ctx* ctx, const float* in, float* out, int64_t n, int64_t P, int mode);
int apply_mask_f32(vipmi_ctx* ctx, const float* in, float* out, int64_t n, int64_t P,
                   const uint8_t* mask, float fill);
int derotate_f32(vipmi_ctx* ctx, const float* in, const double* angles_host, int64_t n, int64_t N,
                 float* out, int mask_nan, int mask_zero, int method, float mask_v = 0.f);
int project_batched_f32(vipmi_ctx* ctx, const float* M, const float* E, int64_t nb, int64_t n, int64_t k, int64_t P,
                        float* R);
int collapse_f32(vipmi_ctx* ctx, const float* cube, int64_t n, int64_t P, int mode, const float* w,
                 int64_t trim_n, float* out);
int collapse_batched_f32(vipmi_ctx* ctx, const float* cube, int64_t batch, int64_t n, int64_t P, int mode, const float* w,
                         int64_t trim_n, float* out);
int gather_f32(vipmi_ctx* ctx, const float* cube, int64_t n, int64_t P, const int32_t* pix,
               int64_t npx, float* A);
int scatter_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t P, const int32_t* pix, int64_t npx,
                float* cube);
int gram_batched_f32(vipmi_ctx* ctx, const float* M, int64_t batch, int64_t n, int64_t P, double* G);
int subset_median_sub_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* idx, const int32_t* len,
                          int64_t wmax, float* out);
int annular_residuals_multi_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* lib_idx,
                                const int32_t* lib_len, int64_t max_lib, const int32_t* ncomps, int64_t nk,
                                float* residuals);
int annular_residuals_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* lib_idx,
                          const int32_t* lib_len, int64_t max_lib, int64_t ncomp, float* residuals);
// small device helpers (util.hip)
int convert_evecs(vipmi_ctx* ctx, const double* evecs, const double* evals, int64_t n, int64_t k,
                  float* Ekn, float* Enk, float* inv_sigma);
// Ct[k][nld] (f32, zero padded) = C[n][k]^T (f64)
int convert_coeffs(vipmi_ctx* ctx, const double* C, int64_t n, int64_t k, float* Ct, int nld);
// out[k][P] = rowscale[c] * in[k][P]
int scale_rows(vipmi_ctx* ctx, const float* in, const float* rowscale, int64_t k, int64_t P, float* out);

}  // namespace vipmi
