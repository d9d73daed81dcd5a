// api.hip -- extern "C" surface of libvipmi.so (see include/vipmi.h), ctx / workspace management and
// the fused full-frame ADI pipeline (psfsub/pca_fullfr.py:801-1007).
#include <stdarg.h>
#include "common.h"
#include "rccl_dl.h"
#include <new>

namespace vipmi {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

}  // namespace vipmi

using namespace vipmi;

int vipmi_ctx::host_scratch(size_t bytes, void** out) {
  if (host_pinned_bytes < bytes) {
    if (host_pinned) {
      hipError_t e = hipStreamSynchronize(stream);          // a read-back into the old block may still be in flight
      if (e != hipSuccess) {
        set_error("host scratch sync failed: %s", hipGetErrorString(e));
        return VIPMI_ERR_HIP;
      }
      (void)hipHostFree(host_pinned);
      host_pinned = nullptr;
      host_pinned_bytes = 0;
    }
    hipError_t e = hipHostMalloc(&host_pinned, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
      host_pinned = nullptr;
      set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
      return VIPMI_ERR_NOMEM;
    }
    host_pinned_bytes = bytes;
  }
  *out = host_pinned;
  return VIPMI_OK;
}

int vipmi_ctx::get(const char* name, size_t bytes, void** out) {
  Buffer& b = buffers[name];
  if (bytes == 0) bytes = 16;
  if (b.bytes < bytes) {
    if (b.ptr) {
      // the old buffer may still be in use by kernels queued on the stream
      hipError_t e = hipStreamSynchronize(stream);
      if (e != hipSuccess) {
        set_error("workspace sync failed: %s", hipGetErrorString(e));
        return VIPMI_ERR_HIP;
      }
      (void)hipFree(b.ptr);
      b.ptr = nullptr;
      b.bytes = 0;
    }
    size_t want = bytes + bytes / 8;     // some slack: avoid re-allocation on small growth
    hipError_t e = hipMalloc(&b.ptr, want);
    if (e != hipSuccess) {
      e = hipMalloc(&b.ptr, bytes);
      want = bytes;
    }
    if (e != hipSuccess) {
      b.ptr = nullptr;
      set_error("workspace '%s': hipMalloc(%zu) failed: %s", name, bytes, hipGetErrorString(e));
      return VIPMI_ERR_NOMEM;
    }
    b.bytes = want;
    upload_keys.erase(name);
  }
  *out = b.ptr;
  return VIPMI_OK;
}

int vipmi_ctx::upload_cached(const char* name, const std::string& key, const void* host, size_t bytes,
                             void** out) {
  void* p = nullptr;
  int s = get(name, bytes, &p);
  if (s != VIPMI_OK) return s;
  auto it = upload_keys.find(name);
  if (it == upload_keys.end() || it->second != key) {
    hipError_t e = hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
      set_error("upload '%s' failed: %s", name, hipGetErrorString(e));
      return VIPMI_ERR_HIP;
    }
    upload_keys[name] = key;
  }
  *out = p;
  return VIPMI_OK;
}

int vipmi_ctx::upload_async(const char* name, const void* host, size_t bytes, void* dst) {
  auto& ring = pinned[name];
  // ONE staging slot by default (option upload_ring): before it is refilled the host waits for the previous upload of this
  // name to have executed, i.e. it runs at most about one call ahead of each stream.  Four slots let it run eight calls ahead
  // of two streams -- no gain (the GPU only needs the next call queued) and a cost: the HIP runtime grows its per-queue pools
  // under the deep queue, and the first ten calls of every process took 6.2 ms instead of 4.9 (tools/pipe_history.py); small
  // cubes: 236 -> 164 us per pipelined call at 50 x 128 x 128 (tools/pipe_small.py).
  if (ring.empty()) ring.resize((size_t)std::min<int64_t>(std::max<int64_t>(opt("upload_ring", 1), 1), 16));
  PinnedSlot& sl = ring[pinned_next[name]++ % ring.size()];
  if (sl.ev) {
    hipError_t e = hipEventSynchronize(sl.ev);
    if (e != hipSuccess) {
      set_error("upload_async: %s", hipGetErrorString(e));
      return VIPMI_ERR_HIP;
    }
  } else if (hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) {
    set_error("upload_async: event creation failed");
    return VIPMI_ERR_HIP;
  }
  if (sl.bytes < bytes) {
    if (sl.host) (void)hipHostFree(sl.host);
    sl.host = nullptr;
    if (hipHostMalloc(&sl.host, bytes + bytes / 4 + 64, hipHostMallocDefault) != hipSuccess) {
      set_error("upload_async: hipHostMalloc(%zu) failed", bytes);
      return VIPMI_ERR_NOMEM;
    }
    sl.bytes = bytes + bytes / 4 + 64;
  }
  memcpy(sl.host, host, bytes);
  hipError_t e = hipMemcpyAsync(dst, sl.host, bytes, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipEventRecord(sl.ev, stream);
  if (e != hipSuccess) {
    set_error("upload_async '%s': %s", name, hipGetErrorString(e));
    return VIPMI_ERR_HIP;
  }
  return VIPMI_OK;
}

int vipmi_ctx::gate_enter() {
  if (!gate || !gate_armed) return VIPMI_OK;
  gate_armed = false;
  std::lock_guard<std::mutex> lk(gate->mu);
  if (gate->last) VIPMI_CHECK_HIP(hipStreamWaitEvent(stream, gate->last, 0));
  return VIPMI_OK;
}

int vipmi_ctx::gate_leave() {
  if (!gate) return VIPMI_OK;
  std::lock_guard<std::mutex> lk(gate->mu);
  if (gate->ring.size() < 64) {
    hipEvent_t e = nullptr;
    VIPMI_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    gate->ring.push_back(e);
    gate->next = (int)gate->ring.size() - 1;
  } else {
    // ring full: the slot about to be re-recorded marks the end of the section issued 64 sections ago.  A stream may still
    // hold a wait on it; the host blocks until that old section has finished (it has, unless more than 64 gated sections
    // are in flight) so that no pending wait ever sees a re-recorded event.
    gate->next = (gate->next + 1) % (int)gate->ring.size();
    VIPMI_CHECK_HIP(hipEventSynchronize(gate->ring[gate->next]));
  }
  hipEvent_t e = gate->ring[gate->next];
  VIPMI_CHECK_HIP(hipEventRecord(e, stream));
  gate->last = e;
  return VIPMI_OK;
}

void vipmi_ctx::tic(const char* stage) {
  if (!timing || (timing == 2 && strcmp(stage, "k_rot_s2") != 0)) return;
  StageTimer& t = timers[stage];
  if (t.open) return;
  if (timing == 3) {
    t.host_t0 = std::chrono::steady_clock::now();
    t.open = true;
    return;
  }
  if (t.used == (int)t.ev.size()) {
    hipEvent_t a = nullptr, b = nullptr;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    t.ev.emplace_back(a, b);
  }
  (void)hipEventRecord(t.ev[t.used].first, stream);
  t.open = true;
}

void vipmi_ctx::toc(const char* stage) {
  if (!timing || (timing == 2 && strcmp(stage, "k_rot_s2") != 0)) return;
  StageTimer& t = timers[stage];
  if (!t.open) return;
  if (timing == 3) {
    t.host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t.host_t0).count();
    t.host_count++;
    t.open = false;
    return;
  }
  (void)hipEventRecord(t.ev[t.used].second, stream);
  t.used++;
  t.open = false;
}

namespace {
// E[b][c][i] (float32) = evecs[b][c][i] (float64 rows, c < k), zeroed where the eigenvalue is below 1e-12 of the
// leading one (components the projection must not use: their vectors are arbitrary)
__global__ void evecs_rows_f32_kernel(const double* __restrict__ evecs, const double* __restrict__ evals, int n, int k,
                                      float* __restrict__ E) {
  const int b = blockIdx.y;
  const double* ev = evals + (size_t)b * n;
  const double lead = ev[0];
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < k * n; e += gridDim.x * blockDim.x) {
    const int c = e / n;
    const bool keep = ev[c] > lead * 1e-12;
    E[(size_t)b * k * n + e] = keep ? (float)evecs[(size_t)b * n * n + e] : 0.f;
  }
}
}  // namespace

extern "C" {

// 105 (round 5): + vipmi_annular_eigh_f64, vipmi_pca_fullframe_f64, vipmi_center_f64, vipmi_gram_offset_f64, vipmi_annular_apply_mu_f32;
//                 recovery of cooperating solves
// 106 (round 5): + vipmi_pca_fullframe_hostin_f32 (the Gram under the upload); float-domain median selection
// 107 (round 6): option sub_guard (the subtraction's zero guard is opt-out per call: median_sub); + vipmi_annular_gram_all_f32,
//                 vipmi_annular_apply_all_f32, vipmi_annular_gram_all_f64 (the fronts of all annulus segments in a handful of launches)
// 108 (round 6): the spatial scalings on the float64 routes: vipmi_pca_fullframe_f64 serves every scaling; + vipmi_spat_center_f64,
//                 vipmi_gram_offset_u_f64, vipmi_annular_apply_mu_u_f32 (the offset u mu^T in the place of 1 mu^T)
int vipmi_version(void) { return 108; }

const char* vipmi_last_error(void) { return g_err; }

int vipmi_create(int device, void* stream, vipmi_ctx** out) {
  VIPMI_REQUIRE(out != nullptr, "vipmi_create: out is null");
  int ndev = 0;
  VIPMI_CHECK_HIP(hipGetDeviceCount(&ndev));
  VIPMI_REQUIRE(device >= 0 && device < ndev, "vipmi_create: device %d out of range (%d devices)", device, ndev);
  VIPMI_CHECK_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  VIPMI_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("libvipmi is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
    return VIPMI_ERR_UNSUPPORTED;
  }
  vipmi_ctx* c = new vipmi_ctx();
  c->device = device;
  c->stream = reinterpret_cast<hipStream_t>(stream);
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  *out = c;
  return VIPMI_OK;
}

int vipmi_destroy(vipmi_ctx* ctx) {
  if (!ctx) return VIPMI_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& kv : ctx->buffers)
    if (kv.second.ptr) (void)hipFree(kv.second.ptr);
  for (auto& kv : ctx->pinned)
    for (auto& sl : kv.second) {
      if (sl.ev) (void)hipEventDestroy(sl.ev);
      if (sl.host) (void)hipHostFree(sl.host);
    }
  for (auto& kv : ctx->timers)
    for (auto& pr : kv.second.ev) {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
  if (ctx->host_pinned) (void)hipHostFree(ctx->host_pinned);
  for (auto e : ctx->copy_events) (void)hipEventDestroy(e);
  if (ctx->copy_stream) {
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamDestroy(ctx->copy_stream);
  }
  delete ctx;
  return VIPMI_OK;
}

int vipmi_trim(vipmi_ctx* ctx) {
  VIPMI_REQUIRE(ctx, "null ctx");
  VIPMI_CHECK_HIP(hipSetDevice(ctx->device));
  VIPMI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  {
    // failures latched on the device and not yet read by vipmi_check_deferred survive the trim on the host (the context may
    // belong to another thread's pipelined region that checks later)
    auto it = ctx->buffers.find("deferred_fail");
    if (it != ctx->buffers.end() && it->second.ptr) {
      int h[4] = {0, 0, 0, 0};
      VIPMI_CHECK_HIP(hipMemcpy(h, it->second.ptr, sizeof h, hipMemcpyDeviceToHost));
      ctx->sticky_fail[0] += h[0];
      ctx->sticky_fail[1] += h[1];
      ctx->options["eigh_recovered"] = ctx->opt("eigh_recovered", 0) + h[3];
    }
  }
  for (auto& kv : ctx->buffers)
    if (kv.second.ptr) (void)hipFree(kv.second.ptr);
  ctx->buffers.clear();
  ctx->upload_keys.clear();
  return VIPMI_OK;
}

int vipmi_set_stream(vipmi_ctx* ctx, void* stream) {
  VIPMI_REQUIRE(ctx, "null ctx");
  ctx->stream = reinterpret_cast<hipStream_t>(stream);
  return VIPMI_OK;
}

int vipmi_synchronize(vipmi_ctx* ctx) {
  VIPMI_REQUIRE(ctx, "null ctx");
  VIPMI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return VIPMI_OK;
}

int vipmi_set_option(vipmi_ctx* ctx, const char* key, int64_t value) {
  VIPMI_REQUIRE(ctx && key, "null argument");
  static const char* known[] = {"timing", "eigh_split", "rot_4096_w1", "ann_large_min", "gram_f32", "gram_tb", "gram_slices", "eigh_max_sweeps",
                                "eigh_check", "rot_ws_mb", "rot_batch", "rot_conv", "reserve_cus", "eigh_method", "eigh_multi", "eigh_nt", "eigh_reg", "eigh_many", "hostin_overlap", "eigh_large_w", "eigh_xl_min", "gram_wpw", "bgemm_tb", "bgemm_lds", "warp_direct", "median_tp", "median_reg", "median_xcd_chunk", "eigh_fast", "eigh_fast_tol", "eigh_fast_budget", "eigh_fast_min", "gram_i8", "gram_i8_slices", "gram_i8_nbuf", "gram_i8_min_n", "gram_i8_dma", "eigh_one_xcd", "eigh_w", "eigh_wave", "eigh_wave_drop", "rot_pair_store", "subtract_lds", "upload_ring", "rot_1024_q", "eigh_recover", "eigh_multi_drop", "eigh_wave_async", "ann_gather", "ann_range", "sub_guard", "eigh_many_jacobi", nullptr};
  bool ok = false;
  for (int i = 0; known[i]; ++i) ok = ok || strcmp(known[i], key) == 0;
  VIPMI_REQUIRE(ok, "unknown option '%s'", key);
  if (strcmp(key, "timing") == 0) ctx->timing = (int)value;
  ctx->options[key] = value;
  return VIPMI_OK;
}

int64_t vipmi_get_option(vipmi_ctx* ctx, const char* key) {
  if (!ctx || !key) return -1;
  return ctx->opt(key, -1);
}

float vipmi_stage_ms(vipmi_ctx* ctx, const char* stage) {
  if (!ctx || !stage) return -1.f;
  auto it = ctx->timers.find(stage);
  if (it != ctx->timers.end() && ctx->timing == 3) return it->second.host_count ? (float)it->second.host_ms : -1.f;
  if (it == ctx->timers.end() || it->second.used == 0) return -1.f;
  StageTimer& t = it->second;
  if (hipEventSynchronize(t.ev[t.used - 1].second) != hipSuccess) return -1.f;
  float total = 0.f;
  for (int i = 0; i < t.used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.ev[i].first, t.ev[i].second) != hipSuccess) return -1.f;
    total += ms;
  }
  return total;
}

int vipmi_stage_count(vipmi_ctx* ctx, const char* stage) {
  if (!ctx || !stage) return -1;
  auto it = ctx->timers.find(stage);
  return it == ctx->timers.end() ? 0 : it->second.used;
}

int vipmi_reset_timers(vipmi_ctx* ctx) {
  VIPMI_REQUIRE(ctx, "null ctx");
  for (auto& kv : ctx->timers) {
    kv.second.used = 0;
    kv.second.open = false;
    kv.second.host_ms = 0.0;
    kv.second.host_count = 0;
  }
  return VIPMI_OK;
}

// Deferred error check for asynchronous use (option "eigh_check"=0): synchronises the stream and reports
// whether any eigensolve since the last check failed to converge.
int vipmi_check_deferred(vipmi_ctx* ctx) {
  VIPMI_REQUIRE(ctx, "null ctx");
  VIPMI_CHECK_HIP(hipSetDevice(ctx->device));
  auto it = ctx->buffers.find("deferred_fail");
  VIPMI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  int h[2] = {ctx->sticky_fail[0], ctx->sticky_fail[1]};
  if (it != ctx->buffers.end() && it->second.ptr) {
    int d[4] = {0, 0, 0, 0};
    VIPMI_CHECK_HIP(hipMemcpy(d, it->second.ptr, sizeof d, hipMemcpyDeviceToHost));
    if (d[0] != 0 || d[1] != 0) VIPMI_CHECK_HIP(hipMemset(it->second.ptr, 0, 2 * sizeof(int)));
    if (d[3] != 0) VIPMI_CHECK_HIP(hipMemset(reinterpret_cast<int*>(it->second.ptr) + 3, 0, sizeof(int)));
    h[0] += d[0];
    h[1] += d[1];
    // cooperating solves that timed out and were solved again by their recovery launch (eigh_tri.hip): not an error, counted
    ctx->options["eigh_recovered"] = ctx->opt("eigh_recovered", 0) + d[3];
  }
  ctx->sticky_fail[0] = ctx->sticky_fail[1] = 0;        // (only once the device words have been read: a failed copy keeps them)
  if (h[1] != 0) {
    set_error("eigh: %d inter-workgroup barrier(s) timed out (cooperating workgroups were not co-resident): the results of "
              "those calls are invalid", h[1]);
    return VIPMI_ERR_HIP;
  }
  if (h[0] != 0) {
    set_error("eigh: %d eigenproblem(s) did not converge", h[0]);
    return VIPMI_ERR_NOCONV;
  }
  return VIPMI_OK;
}

int vipmi_gate_create(vipmi_gate** out) {
  VIPMI_REQUIRE(out, "vipmi_gate_create: out is null");
  *out = new (std::nothrow) vipmi_gate();
  VIPMI_REQUIRE(*out, "vipmi_gate_create: out of memory");
  return VIPMI_OK;
}

int vipmi_gate_destroy(vipmi_gate* gate) {
  if (!gate) return VIPMI_OK;
  for (hipEvent_t e : gate->ring) (void)hipEventDestroy(e);
  delete gate;
  return VIPMI_OK;
}

int vipmi_set_gate(vipmi_ctx* ctx, vipmi_gate* gate) {
  VIPMI_REQUIRE(ctx, "null ctx");
  ctx->gate = gate;
  ctx->gate_armed = false;
  return VIPMI_OK;
}

#define CTX_GUARD()                      \
  VIPMI_REQUIRE(ctx, "null ctx");        \
  VIPMI_CHECK_HIP(hipSetDevice(ctx->device))

int vipmi_scale_f32(vipmi_ctx* ctx, const float* in, float* out, int64_t n, int64_t P, int mode) {
  CTX_GUARD();
  return scale_f32(ctx, in, out, n, P, mode);
}

int vipmi_apply_mask_f32(vipmi_ctx* ctx, const float* in, float* out, int64_t n, int64_t P,
                         const uint8_t* mask, float fill) {
  CTX_GUARD();
  return apply_mask_f32(ctx, in, out, n, P, mask, fill);
}

int vipmi_gram_f32(vipmi_ctx* ctx, const float* M, int64_t n, int64_t P, int64_t ld, double* G) {
  CTX_GUARD();
  return gram_f32(ctx, M, n, M, n, P, ld, G);
}

int vipmi_gram_batched_f32(vipmi_ctx* ctx, const float* M, int64_t batch, int64_t n, int64_t P, double* G) {
  CTX_GUARD();
  return gram_batched_f32(ctx, M, batch, n, P, G);
}

int vipmi_cross_gram_f32(vipmi_ctx* ctx, const float* A, int64_t na, const float* B, int64_t nb,
                         int64_t P, int64_t ld, double* C) {
  CTX_GUARD();
  return gram_f32(ctx, A, na, B, nb, P, ld, C);
}

int vipmi_eigh_topk_f64(vipmi_ctx* ctx, double* G, int64_t batch, int64_t n, int64_t k, const int32_t* nact,
                        double* evals, double* evecs) {
  CTX_GUARD();
  VIPMI_REQUIRE(G && evals && evecs, "eigh_topk: null pointer");
  VIPMI_REQUIRE(batch > 0 && n > 0 && k > 0 && k <= n, "eigh_topk: bad sizes batch=%ld n=%ld k=%ld", (long)batch,
                (long)n, (long)k);
  return eigh_leading(ctx, G, batch, n, k, nact, evals, evecs);
}

int vipmi_lincomb_f32(vipmi_ctx* ctx, const float* x, const float* y, float a, float b, int64_t total, float* out) {
  CTX_GUARD();
  return lincomb_f32(ctx, x, y, a, b, total, out);
}

int vipmi_zoom_frames_f32(vipmi_ctx* ctx, const float* X, int64_t nb, int64_t din, const float* Er, const float* Ei,
                          const int32_t* chan, int64_t dout, int64_t ldk, float* work, float* out) {
  CTX_GUARD();
  VIPMI_REQUIRE(X && Er && Ei && chan && work && out, "zoom_frames: null pointer");
  VIPMI_REQUIRE(nb > 0 && din > 0 && dout > 0 && ldk >= din, "zoom_frames: bad sizes");
  // rows of X that are not 16-byte aligned (din % 4 != 0, e.g. 334-pixel rescaled frames) would push the product kernel
  // onto its scalar-load path (measured 31 instead of 50 TF/s): re-lay X with the operators' row length first
  int64_t ldx = din;
  if (din % 4 != 0 && ldk % 4 == 0) {
    float* xp = nullptr;
    VIPMI_TRY(ws(ctx, "zoom_xpad", (size_t)nb * din * ldk, &xp));
    VIPMI_CHECK_HIP(hipMemcpy2DAsync(xp, (size_t)ldk * sizeof(float), X, (size_t)din * sizeof(float),
                                     (size_t)din * sizeof(float), (size_t)(nb * din), hipMemcpyDeviceToDevice, ctx->stream));
    X = xp;
    ldx = ldk;
  }
  // U = E X^T  (dout x din, per frame, real and imaginary operators) ...
  float* Ur = work;
  float* Ui = work + (size_t)nb * dout * ldk;
  for (int64_t b0 = 0; b0 < nb; b0 += 32768) {
    const int64_t cnt = nb - b0 < 32768 ? nb - b0 : 32768;
    const float* Xb = X + (size_t)b0 * din * ldx;
    VIPMI_TRY(bgemm_abt_f32(ctx, Er, Xb, nullptr, nullptr, chan + b0, nullptr, cnt, dout, din, din, ldk, ldx, ldk,
                            dout * ldk, din * ldx, dout * ldk, Ur + (size_t)b0 * dout * ldk));
    VIPMI_TRY(bgemm_abt_f32(ctx, Ei, Xb, nullptr, nullptr, chan + b0, nullptr, cnt, dout, din, din, ldk, ldx, ldk,
                            dout * ldk, din * ldx, dout * ldk, Ui + (size_t)b0 * dout * ldk));
    // ... Y = Er Ur^T - Ei Ui^T  (dout x dout)
    VIPMI_TRY(bgemm_abt_f32(ctx, Er, Ur + (size_t)b0 * dout * ldk, Ei, Ui + (size_t)b0 * dout * ldk, chan + b0, nullptr,
                            cnt, dout, dout, din, ldk, ldk, dout, dout * ldk, dout * ldk, dout * dout,
                            out + (size_t)b0 * dout * dout));
  }
  return VIPMI_OK;
}

int vipmi_eigh_spectrum_f64(vipmi_ctx* ctx, double* G, int64_t batch, int64_t n, int64_t k, double* evals,
                            double* evecs) {
  CTX_GUARD();
  VIPMI_REQUIRE(G && evals && evecs, "eigh_spectrum: null pointer");
  VIPMI_REQUIRE(batch > 0 && n > 0 && k > 0 && k <= n, "eigh_spectrum: bad sizes batch=%ld n=%ld k=%ld", (long)batch,
                (long)n, (long)k);
  return eigh_leading(ctx, G, batch, n, k, nullptr, evals, evecs, true);
}

int vipmi_eigh_f64(vipmi_ctx* ctx, double* G, int64_t batch, int64_t n, double* evals, double* evecs) {
  CTX_GUARD();
  return eigh_f64(ctx, G, batch, n, evals, evecs);
}

int vipmi_rowspace_gemm_f32(vipmi_ctx* ctx, const float* W, const float* M, int64_t k, int64_t n,
                            int64_t P, const float* rowscale, float* B) {
  CTX_GUARD();
  return rowspace_gemm_f32(ctx, W, M, k, n, P, rowscale, B);
}

int vipmi_subtract_gemm_f32(vipmi_ctx* ctx, const float* M, const float* C, const float* B, int64_t n,
                            int64_t k, int64_t P, float* R, float* recon) {
  CTX_GUARD();
  return subtract_gemm_f32(ctx, M, C, B, n, k, P, R, recon);
}

int vipmi_derotate_f32(vipmi_ctx* ctx, const float* in, const double* angles_host, int64_t n, int64_t N,
                       float* out, int mask_nan, int mask_zero, int method) {
  CTX_GUARD();
  return derotate_f32(ctx, in, angles_host, n, N, out, mask_nan, mask_zero, method);
}

// Only the verified fast path (eigh_chfsi.hip): *converged = 1 -> evals[0..k) / evecs[k][n] hold the leading pairs, each with
// ||G q - theta q|| <= 1e-13 theta_1; 0 -> nothing written (spectrum without a usable gap, or sizes outside 256 <= n <= 16384,
// k + max(12, k/4) <= 64, 4 k <= n).  G is not modified.  What the Python front tries before rocSOLVER above 6144 frames.
int vipmi_eigh_topk_fast_f64(vipmi_ctx* ctx, const double* G, int64_t n, int64_t k, double* evals, double* evecs,
                             int* converged) {
  CTX_GUARD();
  VIPMI_REQUIRE(G && evals && evecs && converged && n > 0 && k > 0, "eigh_topk_fast: bad arguments");
  int info[4];
  StageScope sc(ctx, "eigh");
  VIPMI_TRY(vipmi::eigh_chfsi_f64(ctx, G, n, k, evals, evecs, converged, info));
  ctx->options["eigh_fast_last_products"] = info[0];
  ctx->options["eigh_fast_last_rounds"] = info[1];
  ctx->options["eigh_fast_last_locked"] = info[2];
  ctx->options["eigh_fast_last_reason"] = info[3];
  return VIPMI_OK;
}

int vipmi_derotate_maskval_f32(vipmi_ctx* ctx, const float* in, const double* angles_host, int64_t n, int64_t N,
                               float* out, float mask_val, int method) {
  CTX_GUARD();
  const bool nan = !(mask_val == mask_val);
  return derotate_f32(ctx, in, angles_host, n, N, out, nan ? 1 : 0, nan ? 0 : 1, method, nan ? 0.f : mask_val);
}

int vipmi_rotate_interp_f32(vipmi_ctx* ctx, const float* in, const double* angles_host, int64_t n, int64_t N,
                            double cx, double cy, int interp, int border, float* out) {
  CTX_GUARD();
  return rotate_interp_f32(ctx, in, angles_host, n, N, cx, cy, interp, border, out);
}

int vipmi_collapse_f32(vipmi_ctx* ctx, const float* cube, int64_t n, int64_t P, int mode, const float* w,
                       int64_t trim_n, float* out) {
  CTX_GUARD();
  return collapse_f32(ctx, cube, n, P, mode, w, trim_n, out);
}

int vipmi_project_batched_f32(vipmi_ctx* ctx, const float* M, const float* E, int64_t nb, int64_t n, int64_t k,
                              int64_t P, float* R) {
  CTX_GUARD();
  return project_batched_f32(ctx, M, E, nb, n, k, P, R);
}

int vipmi_collapse_batched_f32(vipmi_ctx* ctx, const float* cubes, int64_t batch, int64_t n, int64_t P, int mode,
                               const float* w, int64_t trim_n, float* out) {
  CTX_GUARD();
  return collapse_batched_f32(ctx, cubes, batch, n, P, mode, w, trim_n, out);
}

int vipmi_subset_median_sub_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* lib_idx,
                                const int32_t* lib_len, int64_t max_lib, float* out) {
  CTX_GUARD();
  return subset_median_sub_f32(ctx, A, n, npx, lib_idx, lib_len, max_lib, out);
}

int vipmi_gather_f32(vipmi_ctx* ctx, const float* cube, int64_t n, int64_t P, const int32_t* pix,
                     int64_t npx, float* A) {
  CTX_GUARD();
  return gather_f32(ctx, cube, n, P, pix, npx, A);
}

int vipmi_scatter_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t P, const int32_t* pix, int64_t npx,
                      float* cube) {
  CTX_GUARD();
  return scatter_f32(ctx, A, n, P, pix, npx, cube);
}

int vipmi_annular_residuals_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx,
                                const int32_t* lib_idx, const int32_t* lib_len, int64_t max_lib,
                                int64_t ncomp, float* residuals) {
  CTX_GUARD();
  return annular_residuals_f32(ctx, A, n, npx, lib_idx, lib_len, max_lib, ncomp, residuals);
}

int vipmi_annular_subgrams_f64(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* lib_idx,
                               const int32_t* lib_len, int64_t max_lib, int64_t m, double* G, double* H) {
  VIPMI_REQUIRE(ctx, "null ctx");
  return annular_subgrams_f64(ctx, A, n, npx, lib_idx, lib_len, max_lib, m, G, H);
}

int vipmi_annular_eigh_f64(vipmi_ctx* ctx, const double* G, int64_t nseg, int64_t n, const int32_t* lib_idx,
                           const int32_t* lib_len, int64_t m, int64_t k, double* work, double* evals, double* evecs) {
  CTX_GUARD();
  return annular_eigh_f64(ctx, G, nseg, n, lib_idx, lib_len, m, k, work, evals, evecs);
}

int vipmi_annular_gram_all_f32(vipmi_ctx* ctx, const float* cube, int64_t n, int64_t P, const int32_t* pix_all, int64_t Ptot,
                               int64_t klen, const int32_t* seg_slice, int64_t nseg, float* A_all, double* G_all) {
  CTX_GUARD();
  return annular_gram_all_f32(ctx, cube, n, P, pix_all, Ptot, klen, seg_slice, nseg, A_all, G_all);
}

int vipmi_annular_apply_all_f32(vipmi_ctx* ctx, const float* A_all, int64_t n, int64_t Ptot, const int32_t* tile_seg,
                                const int32_t* pix_out, int64_t nseg, const int32_t* lib_idx, const int32_t* lib_len, int64_t m,
                                const double* G_all, const double* evals, const double* evecs, const int32_t* kseg, int64_t kmax,
                                int64_t P, float* cube_out, const float* mu32) {
  CTX_GUARD();
  return annular_apply_all_f32(ctx, A_all, n, Ptot, tile_seg, pix_out, nseg, lib_idx, lib_len, m, G_all, evals, evecs, kseg, kmax, P,
                               cube_out, mu32);
}

int vipmi_annular_gram_all_f64(vipmi_ctx* ctx, const double* cube, int64_t n, int64_t P, const int32_t* pix_all, int64_t Ptot,
                               int64_t klen, const int32_t* seg_slice, int64_t nseg, int mode, float* D_all, double* mu, float* mu32,
                               double* G_all) {
  CTX_GUARD();
  return annular_gram_all_f64(ctx, cube, n, P, pix_all, Ptot, klen, seg_slice, nseg, mode, D_all, mu, mu32, G_all);
}

int vipmi_annular_apply_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* lib_idx,
                            const int32_t* lib_len, int64_t max_lib, int64_t m, const double* G, const double* evals,
                            const double* evecs, const int32_t* ncomps_host, int64_t nk, float* residuals) {
  VIPMI_REQUIRE(ctx, "null ctx");
  return annular_apply_f32(ctx, A, n, npx, lib_idx, lib_len, max_lib, m, G, evals, evecs, ncomps_host, nk, residuals);
}

int vipmi_annular_residuals_multi_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx,
                                      const int32_t* lib_idx, const int32_t* lib_len, int64_t max_lib,
                                      const int32_t* ncomps_host, int64_t nk, float* residuals) {
  CTX_GUARD();
  return annular_residuals_multi_f32(ctx, A, n, npx, lib_idx, lib_len, max_lib, ncomps_host, nk, residuals);
}

// PCs / residuals of M[n,P] with respect to the top-k principal components of ref[nref,P]
// (ref == M for ADI).  Internal building block of vipmi_pca_fullframe_f32 and of the RDI path.
int vipmi_pca_project_f32(vipmi_ctx* ctx, const float* M, int64_t n, const float* ref, int64_t nref,
                          int64_t P, int64_t k, float* residuals, float* recon, float* pcs,
                          double* evals_out) {
  CTX_GUARD();
  VIPMI_REQUIRE(M && ref && residuals, "pca_project: null pointer");
  VIPMI_REQUIRE(k > 0 && k <= nref && k <= P, "%ld PCs cannot be obtained from a matrix with size [%ld,%ld].",
                (long)k, (long)nref, (long)P);
  double *G = nullptr, *evals = nullptr, *evecs = nullptr;
  VIPMI_TRY(ws(ctx, "pca_G", (size_t)nref * nref, &G));
  VIPMI_TRY(ws(ctx, "pca_evals", (size_t)nref, &evals));
  VIPMI_TRY(ws(ctx, "pca_evecs", (size_t)nref * nref, &evecs));
  if (ctx->gram_given_ref == ref && ctx->gram_given_n == nref) {
    ctx->gram_given_ref = nullptr;                 // the host-input front formed ref ref^T while the rows arrived
  } else {
    ctx->gram_given_ref = nullptr;
    VIPMI_TRY(gram_f32(ctx, ref, nref, ref, nref, P, P, G));
  }
  // evals_out: the caller wants the whole spectrum (values only beyond the k leading pairs)
  VIPMI_TRY(eigh_leading(ctx, G, 1, nref, k, nullptr, evals, evecs, evals_out != nullptr));
  VIPMI_TRY(ctx->gate_enter());
  const int nld = (int)cdiv(nref, 32) * 32, kld = (int)cdiv(k, 32) * 32;
  float *Ekn = nullptr, *Enk = nullptr, *isig = nullptr;
  VIPMI_TRY(ws(ctx, "pca_Ekn", (size_t)kld * nld, &Ekn));
  VIPMI_TRY(ws(ctx, "pca_Enk", (size_t)nld * kld, &Enk));
  VIPMI_TRY(ws(ctx, "pca_isig", (size_t)kld, &isig));
  VIPMI_TRY(convert_evecs(ctx, evecs, evals, nref, k, Ekn, Enk, isig));
  if (evals_out)
    VIPMI_CHECK_HIP(hipMemcpyAsync(evals_out, evals, sizeof(double) * nref, hipMemcpyDeviceToDevice, ctx->stream));
  float* T = nullptr;
  VIPMI_TRY(ws(ctx, "pca_T", (size_t)k * P, &T));
  {
    StageScope sc(ctx, "project");
    if (ref == M && nref == n) {
      // ADI: reconstructed = E (E^T M)
      VIPMI_TRY(rowspace_gemm_t(ctx, Enk, kld, M, k, n, P, nullptr, T));
      VIPMI_TRY(subtract_gemm_t(ctx, M, Ekn, nld, T, n, k, P, residuals, recon));
      if (pcs) VIPMI_TRY(scale_rows(ctx, T, isig, k, P, pcs));     // V = S^-1 E^T M
    } else {
      // RDI: V = S^-1 E^T ref ; coefficients C = M V^T (n x k) ; reconstructed = C V
      float* V = pcs ? pcs : T;
      VIPMI_TRY(rowspace_gemm_t(ctx, Enk, kld, ref, k, nref, P, isig, V));
      double* C64 = nullptr;
      VIPMI_TRY(ws(ctx, "pca_C64", (size_t)n * k, &C64));
      VIPMI_TRY(gram_f32(ctx, M, n, V, k, P, P, C64));
      // C (n x k, f64) -> Ct [k][nld2] f32
      const int nld2 = (int)cdiv(n, 32) * 32;
      float* Ct = nullptr;
      VIPMI_TRY(ws(ctx, "pca_Ct", (size_t)kld * nld2, &Ct));
      VIPMI_TRY(convert_coeffs(ctx, C64, n, k, Ct, nld2));
      VIPMI_TRY(subtract_gemm_t(ctx, M, Ct, nld2, V, n, k, P, residuals, recon));
    }
  }
  return VIPMI_OK;
}

int vipmi_pca_fullframe_f32(vipmi_ctx* ctx, const float* cube, const double* angles_host, int64_t n,
                            int64_t N, int64_t ncomp, int scaling, const uint8_t* mask, int collapse_mode,
                            float* frame, float* pcs, float* recon, float* residuals, float* residuals_der) {
  CTX_GUARD();
  VIPMI_REQUIRE(cube && angles_host && frame, "pca_fullframe: null pointer");
  VIPMI_REQUIRE(n > 0 && N > 1, "pca_fullframe: bad sizes");
  VIPMI_REQUIRE(ncomp > 0, "Number of PCs too low. It should be > 0.");
  const int64_t P = N * N;
  int64_t k = ncomp > n ? n : ncomp;            // pca_fullfr.py:876-881 (clamp, not an error)
  const float* M = cube;
  float* scratch = nullptr;
  if (mask || scaling) VIPMI_TRY(ws(ctx, "pca_M", (size_t)n * P, &scratch));
  if (mask) {
    VIPMI_TRY(apply_mask_f32(ctx, M, scratch, n, P, mask, 0.f));
    M = scratch;
  }
  if (scaling) {
    VIPMI_TRY(scale_f32(ctx, M, scratch, n, P, scaling));
    M = scratch;
  }
  float* res = residuals;
  if (!res) VIPMI_TRY(ws(ctx, "pca_res", (size_t)n * P, &res));
  ctx->gate_armed = ctx->gate != nullptr;
  VIPMI_TRY(vipmi_pca_project_f32(ctx, M, n, M, n, P, k, res, recon, pcs, nullptr));
  float* der = residuals_der;
  if (!der) VIPMI_TRY(ws(ctx, "pca_der", (size_t)n * P, &der));
  // pca(): mask_center_px without rot_options -> mask_val=0 (pca_fullfr.py:412-415)
  VIPMI_TRY(derotate_f32(ctx, res, angles_host, n, N, der, mask ? 0 : 1, mask ? 1 : 0, VIPMI_ROT_AUTO));
  VIPMI_TRY(collapse_f32(ctx, der, n, P, collapse_mode, nullptr, 50, frame));   // cube_collapse(..., n=50) default (subsampling.py:30)
  VIPMI_TRY(ctx->gate_leave());
  if (mask) {
    // pca_fullfr.py:985-987: residuals_cube_ and frame are masked again
    if (residuals_der) VIPMI_TRY(apply_mask_f32(ctx, der, der, n, P, mask, 0.f));
    VIPMI_TRY(apply_mask_f32(ctx, frame, frame, 1, P, mask, 0.f));
  }
  return VIPMI_OK;
}

// The fused call for a cube that is still in HOST memory (a caller's numpy array: psfsub/pca_fullfr.py:137 takes nothing else).
// The upload of a C2 cube takes longer than the whole PCA (8.0 against 6.2 ms) and nothing can start before its last frame --
// except the Gram matrix, whose tile (i, j) needs only the row blocks i and j: the cube is copied in blocks of 64 frames on a
// copy stream, and behind every block the compute stream splits it into digit planes and multiplies it with the blocks that are
// already there (gram_i8_inc_*).  After the last block a quarter of the product and the reduction are left: ~0.5 of the 0.7 ms of
// the Gram stage disappear under the copy.  Bit-identical to vipmi_pca_fullframe_f32 on the uploaded cube (same partial sums, same
// order).  cube: device buffer [n][N][N] that receives the copy (the caller keeps it: residuals etc. refer to it).
int vipmi_pca_fullframe_hostin_f32(vipmi_ctx* ctx, const float* host_cube, float* cube, const double* angles_host, int64_t n, int64_t N,
                                   int64_t ncomp, const uint8_t* mask, int collapse_mode, float* frame, float* pcs, float* recon,
                                   float* residuals, float* residuals_der) {
  CTX_GUARD();
  VIPMI_REQUIRE(host_cube && cube && angles_host && frame, "pca_fullframe_hostin: null pointer");
  VIPMI_REQUIRE(n > 0 && N > 1, "pca_fullframe_hostin: bad sizes");
  const int64_t P = N * N;
  const bool overlap = gram_i8_default_path(ctx, n, P) && ctx->opt("hostin_overlap", 1) != 0 && ncomp > 0 &&
                       (reinterpret_cast<uintptr_t>(cube) & 15) == 0 && (P & 3) == 0;
  auto plain = [&]() -> int {
    VIPMI_CHECK_HIP(hipMemcpyAsync(cube, host_cube, sizeof(float) * (size_t)n * P, hipMemcpyHostToDevice, ctx->stream));
    return vipmi_pca_fullframe_f32(ctx, cube, angles_host, n, N, ncomp, 0, mask, collapse_mode, frame, pcs, recon, residuals,
                                   residuals_der);
  };
  if (!overlap) return plain();
  if (!ctx->copy_stream) VIPMI_CHECK_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  GramI8Inc st;
  {
    // the digit planes take 5 bytes per sample on top of the cube: when they do not fit, upload first and let gram_f32 choose
    // (it falls back to the float64-MFMA kernel on its own, gram.hip) -- a cube that worked through upload + pca must not fail here
    const int rc0 = gram_i8_inc_begin(ctx, n, P, P, &st);
    if (rc0 == VIPMI_ERR_NOMEM) {
      (void)hipGetLastError();
      return plain();
    }
    VIPMI_TRY(rc0);
  }
  while ((int)ctx->copy_events.size() < st.nt + 1) {
    hipEvent_t e = nullptr;
    VIPMI_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->copy_events.push_back(e);
  }
  double* G = nullptr;
  VIPMI_TRY(ws(ctx, "pca_G", (size_t)n * n, &G));
  // mask_center_px: the decomposition is that of the MASKED matrix (pca_fullfr.py:1627-1630); the mask is per pixel, so every
  // block is masked as it arrives, into the workspace the fused call below masks the whole cube into again (same values)
  float* masked = nullptr;
  if (mask) VIPMI_TRY(ws(ctx, "pca_M", (size_t)n * P, &masked));
  // the copy stream starts behind whatever the compute stream has queued (the destination may still be in use there)
  VIPMI_CHECK_HIP(hipEventRecord(ctx->copy_events[st.nt], ctx->stream));
  VIPMI_CHECK_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->copy_events[st.nt], 0));
  for (int b = 0; b < st.nt; ++b) {
    const int64_t r0 = (int64_t)64 * b, r1 = r0 + 64 < n ? r0 + 64 : n;
    if (r1 > r0)
      VIPMI_CHECK_HIP(hipMemcpyAsync(cube + r0 * P, host_cube + r0 * P, sizeof(float) * (size_t)(r1 - r0) * P, hipMemcpyHostToDevice,
                                     ctx->copy_stream));
    VIPMI_CHECK_HIP(hipEventRecord(ctx->copy_events[b], ctx->copy_stream));
    VIPMI_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->copy_events[b], 0));
    if (mask && r1 > r0) VIPMI_TRY(apply_mask_f32(ctx, cube + r0 * P, masked + r0 * P, r1 - r0, P, mask, 0.f));
    VIPMI_TRY(gram_i8_inc_block(ctx, st, mask ? masked : cube, b));
  }
  VIPMI_TRY(gram_i8_inc_end(ctx, st, G));
  ctx->gram_given_ref = mask ? masked : cube;
  ctx->gram_given_n = n;
  const int rc = vipmi_pca_fullframe_f32(ctx, cube, angles_host, n, N, ncomp, 0, mask, collapse_mode, frame, pcs, recon, residuals,
                                         residuals_der);
  ctx->gram_given_ref = nullptr;
  return rc;
}

int vipmi_center_f64(vipmi_ctx* ctx, const double* M, int64_t n, int64_t P, int mode, float* D, double* mu, float* mu32) {
  CTX_GUARD();
  return center_f64(ctx, M, n, P, mode, D, mu, mu32);
}

int vipmi_gram_offset_f64(vipmi_ctx* ctx, const float* D, const double* mu, int64_t n, int64_t P, double* G) {
  CTX_GUARD();
  return gram_offset_f64(ctx, D, mu, n, P, G);
}

int vipmi_spat_center_f64(vipmi_ctx* ctx, const double* M, int64_t n, int64_t P, int64_t Preal, int with_std, float* D, double* mu,
                          float* mu32, double* u) {
  CTX_GUARD();
  return spat_center_f64(ctx, M, n, P, Preal, with_std, D, mu, mu32, u);
}

int vipmi_gram_offset_u_f64(vipmi_ctx* ctx, const float* D, const double* mu, const double* u, int64_t n, int64_t P, double* G) {
  CTX_GUARD();
  return gram_offset_f64(ctx, D, mu, n, P, G, u);
}

int vipmi_annular_apply_mu_u_f32(vipmi_ctx* ctx, const float* D, int64_t n, int64_t npx, const int32_t* lib_idx, const int32_t* lib_len,
                                 int64_t max_lib, int64_t m, const double* G, const double* evals, const double* evecs,
                                 const int32_t* ncomps_host, int64_t nk, const float* mu32, const double* u, float* residuals) {
  CTX_GUARD();
  return annular_apply_f32(ctx, D, n, npx, lib_idx, lib_len, max_lib, m, G, evals, evecs, ncomps_host, nk, residuals, mu32, u);
}

int vipmi_annular_apply_mu_f32(vipmi_ctx* ctx, const float* D, int64_t n, int64_t npx, const int32_t* lib_idx, const int32_t* lib_len,
                               int64_t max_lib, int64_t m, const double* G, const double* evals, const double* evecs,
                               const int32_t* ncomps_host, int64_t nk, const float* mu32, float* residuals) {
  CTX_GUARD();
  return annular_apply_f32(ctx, D, n, npx, lib_idx, lib_len, max_lib, m, G, evals, evecs, ncomps_host, nk, residuals, mu32);
}

int vipmi_pca_fullframe_f64(vipmi_ctx* ctx, const double* cube, const double* angles_host, int64_t n, int64_t N, int64_t ncomp,
                            int scaling, const uint8_t* mask, int collapse_mode, float* frame, float* pcs, float* recon,
                            float* residuals, float* residuals_der) {
  CTX_GUARD();
  return pca_fullframe_f64(ctx, cube, angles_host, n, N, ncomp, scaling, mask, collapse_mode, frame, pcs, recon, residuals, residuals_der);
}

// 4-D cube without scale_list (psfsub/pca_fullfr.py:544-658): one full-frame ADI PCA per spectral channel with the
// small stages batched -- ONE Gram launch and ONE eigensolver launch for all channels (a workgroup per channel), the
// projections of all channels in two launches, ONE derotation over all nch * n residual frames, ONE collapse launch --
// then the spectral collapse of the per-channel frames.
int vipmi_pca_4d_f32(vipmi_ctx* ctx, const float* cube4, const double* angles_host, int64_t nch, int64_t n, int64_t N,
                     int64_t ncomp, int scaling, const uint8_t* mask, int collapse_mode, int collapse_ifs_mode,
                     float* frame, float* ifs_frames) {
  CTX_GUARD();
  VIPMI_REQUIRE(cube4 && angles_host && frame, "pca_4d: null pointer");
  VIPMI_REQUIRE(nch > 0 && n > 0 && N > 1, "pca_4d: bad sizes");
  VIPMI_REQUIRE(ncomp > 0, "Number of PCs too low. It should be > 0.");
  VIPMI_REQUIRE(collapse_mode != VIPMI_COLLAPSE_WMEAN && collapse_ifs_mode != VIPMI_COLLAPSE_WMEAN,
                "pca_4d: weighted collapses need weights (use vipmi_collapse_f32 on the per-channel frames)");
  const int64_t P = N * N;
  const int64_t k = ncomp > n ? n : ncomp;          // pca_fullfr.py:876-881 (clamp, not an error)
  VIPMI_REQUIRE(n <= 512 && k <= 64, "pca_4d: the batched eigensolver takes up to 512 frames and 64 PCs per channel "
                "(got %ld, %ld): call vipmi_pca_fullframe_f32 per channel", (long)n, (long)k);
  const float* M = cube4;
  if (mask || scaling) {
    float* scratch = nullptr;
    VIPMI_TRY(ws(ctx, "pca4_M", (size_t)nch * n * P, &scratch));
    for (int64_t c = 0; c < nch; ++c) {
      const float* src = cube4 + (size_t)c * n * P;
      float* dst = scratch + (size_t)c * n * P;
      if (mask) {
        VIPMI_TRY(apply_mask_f32(ctx, src, dst, n, P, mask, 0.f));
        src = dst;
      }
      if (scaling) VIPMI_TRY(scale_f32(ctx, src, dst, n, P, scaling));
    }
    M = scratch;
  }
  double *G = nullptr, *evals = nullptr, *evecs = nullptr;
  VIPMI_TRY(ws(ctx, "pca4_G", (size_t)nch * n * n, &G));
  VIPMI_TRY(ws(ctx, "pca4_evals", (size_t)nch * n, &evals));
  VIPMI_TRY(ws(ctx, "pca4_evecs", (size_t)nch * n * n, &evecs));
  VIPMI_TRY(gram_batched_f32(ctx, M, nch, n, P, G));
  VIPMI_TRY(eigh_leading(ctx, G, nch, n, k, nullptr, evals, evecs));
  float* E = nullptr;
  VIPMI_TRY(ws(ctx, "pca4_E", (size_t)nch * k * n, &E));
  {
    const unsigned gx = (unsigned)cdiv(k * n, 256);
    hipLaunchKernelGGL(evecs_rows_f32_kernel, dim3(gx > 64 ? 64 : gx, (unsigned)nch), dim3(256), 0, ctx->stream, evecs, evals,
                       (int)n, (int)k, E);
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  float *R = nullptr, *der = nullptr;
  VIPMI_TRY(ws(ctx, "pca4_res", (size_t)nch * n * P, &R));
  VIPMI_TRY(ws(ctx, "pca4_der", (size_t)nch * n * P, &der));
  VIPMI_TRY(project_batched_f32(ctx, M, E, nch, n, k, P, R));
  static thread_local std::vector<double> tiled;
  tiled.resize((size_t)nch * n);
  for (int64_t c = 0; c < nch; ++c) memcpy(tiled.data() + (size_t)c * n, angles_host, sizeof(double) * n);
  VIPMI_TRY(derotate_f32(ctx, R, tiled.data(), nch * n, N, der, mask ? 0 : 1, mask ? 1 : 0, VIPMI_ROT_AUTO));
  float* ifs = ifs_frames;
  if (!ifs) VIPMI_TRY(ws(ctx, "pca4_ifs", (size_t)nch * P, &ifs));
  VIPMI_TRY(collapse_batched_f32(ctx, der, nch, n, P, collapse_mode, nullptr, 50, ifs));
  if (mask) VIPMI_TRY(apply_mask_f32(ctx, ifs, ifs, nch, P, mask, 0.f));     // pca_fullfr.py:985-987 per channel
  VIPMI_TRY(collapse_f32(ctx, ifs, nch, P, collapse_ifs_mode, nullptr, 50, frame));
  return VIPMI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-GPU variant of vipmi_pca_fullframe_f32 over an RCCL communicator: ONE cube sharded over all ranks with the data
// path collectives of SURVEY 8(e) row "C2/C5 single cube" and nothing else --
//   pixel rows sharded -> partial Gram -> all-reduce (n x n float64) -> identical leading eigenvectors on every rank ->
//   local project / subtract -> all-to-all (row slabs -> whole frames, frames sharded) -> local derotation -> all-to-all
//   back (-> row slabs of all frames) -> local collapse -> exchange of the final frame's row slabs.
// (vip_amd/dist.py:pca_single_cube is the same partition over torch.distributed.)
#define VIPMI_CHECK_RCCL(expr)                                                                                  \
  do {                                                                                                          \
    ncclResult_t _r = (expr);                                                                                   \
    if (_r != ncclSuccess) {                                                                                    \
      vipmi::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, vipmi::rccl_api().GetErrorString(_r));     \
      return VIPMI_ERR_HIP;                                                                                     \
    }                                                                                                           \
  } while (0)

// inside an ncclGroupStart / ncclGroupEnd pair: close the group before returning, or the communicator stays in group mode
#define VIPMI_CHECK_RCCL_IN_GROUP(expr)                                                                         \
  do {                                                                                                          \
    ncclResult_t _r = (expr);                                                                                   \
    if (_r != ncclSuccess) {                                                                                    \
      (void)vipmi::rccl_api().GroupEnd();                                                                       \
      vipmi::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, vipmi::rccl_api().GetErrorString(_r));     \
      return VIPMI_ERR_HIP;                                                                                     \
    }                                                                                                           \
  } while (0)

static inline int64_t split_edge(int64_t total, int world, int r) {      // contiguous near-equal blocks (dist._split)
  const int64_t base = total / world, rem = total % world;
  return r * base + (r < rem ? r : rem);
}

int vipmi_rccl_load(const char* path) {
  const char* why = "";
  if (!vipmi::rccl_load(path, &why)) {
    vipmi::set_error("RCCL could not be loaded: %s", why ? why : "?");
    return VIPMI_ERR_UNSUPPORTED;
  }
  return VIPMI_OK;
}

int vipmi_rccl_unique_id(void* id128) {
  VIPMI_REQUIRE(id128, "rccl_unique_id: null pointer");
  VIPMI_TRY(vipmi_rccl_load(nullptr));
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  VIPMI_CHECK_RCCL(vipmi::rccl_api().GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128)));
  return VIPMI_OK;
}

int vipmi_rccl_comm_create(vipmi_ctx* ctx, const void* id128, int rank, int world, void** comm) {
  CTX_GUARD();
  VIPMI_REQUIRE(id128 && comm && world > 0 && rank >= 0 && rank < world, "rccl_comm_create: bad arguments");
  VIPMI_TRY(vipmi_rccl_load(nullptr));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  VIPMI_CHECK_RCCL(vipmi::rccl_api().CommInitRank(&c, world, id, rank));
  *comm = c;
  return VIPMI_OK;
}

int vipmi_rccl_comm_destroy(void* comm) {
  if (!comm) return VIPMI_OK;
  VIPMI_REQUIRE(vipmi::rccl_api().ok(), "rccl_comm_destroy: RCCL is not loaded");
  VIPMI_CHECK_RCCL(vipmi::rccl_api().CommDestroy(static_cast<ncclComm_t>(comm)));
  return VIPMI_OK;
}

int vipmi_pca_fullframe_sharded_f32(vipmi_ctx* ctx, void* comm_, int rank, int world, const float* slab,
                                    const double* angles_host, int64_t n, int64_t N, int64_t ncomp, int collapse_mode,
                                    float* frame) {
  CTX_GUARD();
  VIPMI_REQUIRE(comm_ && slab && angles_host && frame, "pca_fullframe_sharded: null pointer");
  VIPMI_REQUIRE(world > 0 && rank >= 0 && rank < world && n > 0 && N > 1, "pca_fullframe_sharded: bad sizes");
  VIPMI_REQUIRE(ncomp > 0, "Number of PCs too low. It should be > 0.");
  VIPMI_REQUIRE(collapse_mode != VIPMI_COLLAPSE_WMEAN, "pca_fullframe_sharded: weighted collapse is not offered");
  VIPMI_REQUIRE(vipmi::rccl_api().ok(), "pca_fullframe_sharded: RCCL is not loaded (vipmi_rccl_comm_create first)");
  const vipmi::RcclApi& nc = vipmi::rccl_api();
  ncclComm_t comm = static_cast<ncclComm_t>(comm_);
  hipStream_t st = ctx->stream;
  const int64_t k = ncomp > n ? n : ncomp;
  // every size limit is checked BEFORE the first collective is enqueued (a rank that returned early would leave its peers
  // waiting in the all-reduce): the hand-written eigensolvers stop at 6144 frames
  VIPMI_REQUIRE(vipmi::eigh_topk_supported(n, k) || vipmi::eigh_large_supported(n, k) || n <= 2048,
                "pca_fullframe_sharded: %lld frames are beyond the device eigensolvers", (long long)n);
  const int64_t y0 = split_edge(N, world, rank), y1 = split_edge(N, world, rank + 1), rl = y1 - y0;   // my pixel rows
  const int64_t f0 = split_edge(n, world, rank), f1 = split_edge(n, world, rank + 1), fl = f1 - f0;   // my frames
  const int64_t Pl = rl * N;
  // 1. partial Gram of the slab [n][rl * N], all-reduce
  double *G = nullptr, *evals = nullptr, *evecs = nullptr;
  VIPMI_TRY(ws(ctx, "shard_G", (size_t)n * n, &G));
  VIPMI_TRY(ws(ctx, "shard_evals", (size_t)n, &evals));
  VIPMI_TRY(ws(ctx, "shard_evecs", (size_t)n * n, &evecs));
  if (Pl > 0) {
    VIPMI_TRY(gram_batched_f32(ctx, slab, 1, n, Pl, G));
  } else {
    VIPMI_CHECK_HIP(hipMemsetAsync(G, 0, sizeof(double) * n * n, st));
  }
  VIPMI_CHECK_RCCL(nc.AllReduce(G, G, (size_t)n * n, ncclDouble, ncclSum, comm, st));
  // 2. identical decomposition on every rank, residuals of the slab
  VIPMI_TRY(eigh_leading(ctx, G, 1, n, k, nullptr, evals, evecs));
  float *E = nullptr, *R = nullptr;
  VIPMI_TRY(ws(ctx, "shard_E", (size_t)k * n, &E));
  VIPMI_TRY(ws(ctx, "shard_R", (size_t)n * (Pl > 0 ? Pl : 1), &R));
  {
    const unsigned gx = (unsigned)cdiv(k * n, 256);
    hipLaunchKernelGGL(evecs_rows_f32_kernel, dim3(gx > 64 ? 64 : gx, 1), dim3(256), 0, st, evecs, evals, (int)n, (int)k, E);
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  if (Pl > 0) VIPMI_TRY(project_batched_f32(ctx, slab, E, 1, n, k, Pl, R));
  // 3. row slabs -> whole frames of my frame shard: peer p gets frames [fa, fb) of my slab (one contiguous block) and
  //    sends its rows of my frames, placed into the frames with a strided copy
  float *F = nullptr, *D = nullptr, *stage = nullptr;
  const int64_t rl_max = cdiv(N, world), fl_max = cdiv(n, world);
  VIPMI_TRY(ws(ctx, "shard_F", (size_t)(fl > 0 ? fl : 1) * N * N, &F));
  VIPMI_TRY(ws(ctx, "shard_D", (size_t)(fl > 0 ? fl : 1) * N * N, &D));
  VIPMI_TRY(ws(ctx, "shard_stage", (size_t)world * fl_max * rl_max * N, &stage));
  const size_t chunk = (size_t)fl_max * rl_max * N;            // staging area per peer
  VIPMI_CHECK_RCCL(nc.GroupStart());
  for (int p = 0; p < world; ++p) {
    const int64_t fa = split_edge(n, world, p), fb = split_edge(n, world, p + 1);
    const int64_t ra = split_edge(N, world, p), rb = split_edge(N, world, p + 1);
    if ((fb - fa) * Pl > 0) VIPMI_CHECK_RCCL_IN_GROUP(nc.Send(R + (size_t)fa * Pl, (size_t)(fb - fa) * Pl, ncclFloat, p, comm, st));
    if (fl * (rb - ra) > 0) VIPMI_CHECK_RCCL_IN_GROUP(nc.Recv(stage + p * chunk, (size_t)fl * (rb - ra) * N, ncclFloat, p, comm, st));
  }
  VIPMI_CHECK_RCCL(nc.GroupEnd());
  for (int p = 0; p < world && fl > 0; ++p) {
    const int64_t ra = split_edge(N, world, p), rb = split_edge(N, world, p + 1);
    if (rb > ra)
      VIPMI_CHECK_HIP(hipMemcpy2DAsync(F + (size_t)ra * N, sizeof(float) * N * N, stage + p * chunk,
                                       sizeof(float) * (rb - ra) * N, sizeof(float) * (rb - ra) * N, (size_t)fl,
                                       hipMemcpyDeviceToDevice, st));
  }
  // 4. derotate my frames
  if (fl > 0) VIPMI_TRY(derotate_f32(ctx, F, angles_host + f0, fl, N, D, 1, 0, VIPMI_ROT_AUTO));
  // 5. whole frames -> row slabs of ALL frames: peer p gets its rows of my frames (packed), I get my rows of its frames
  float* S = nullptr;
  VIPMI_TRY(ws(ctx, "shard_S", (size_t)n * (Pl > 0 ? Pl : 1), &S));
  for (int p = 0; p < world && fl > 0; ++p) {
    const int64_t ra = split_edge(N, world, p), rb = split_edge(N, world, p + 1);
    if (rb > ra)
      VIPMI_CHECK_HIP(hipMemcpy2DAsync(stage + p * chunk, sizeof(float) * (rb - ra) * N, D + (size_t)ra * N,
                                       sizeof(float) * N * N, sizeof(float) * (rb - ra) * N, (size_t)fl,
                                       hipMemcpyDeviceToDevice, st));
  }
  VIPMI_CHECK_RCCL(nc.GroupStart());
  for (int p = 0; p < world; ++p) {
    const int64_t fa = split_edge(n, world, p), fb = split_edge(n, world, p + 1);
    const int64_t ra = split_edge(N, world, p), rb = split_edge(N, world, p + 1);
    if (fl * (rb - ra) > 0) VIPMI_CHECK_RCCL_IN_GROUP(nc.Send(stage + p * chunk, (size_t)fl * (rb - ra) * N, ncclFloat, p, comm, st));
    if ((fb - fa) * Pl > 0) VIPMI_CHECK_RCCL_IN_GROUP(nc.Recv(S + (size_t)fa * Pl, (size_t)(fb - fa) * Pl, ncclFloat, p, comm, st));
  }
  VIPMI_CHECK_RCCL(nc.GroupEnd());
  // 6. collapse my pixels over all frames, then every rank collects the row slabs of the final frame
  if (Pl > 0) VIPMI_TRY(collapse_f32(ctx, S, n, Pl, collapse_mode, nullptr, 50, frame + (size_t)y0 * N));
  VIPMI_CHECK_RCCL(nc.GroupStart());
  for (int p = 0; p < world; ++p) {
    if (p == rank) continue;
    const int64_t ra = split_edge(N, world, p), rb = split_edge(N, world, p + 1);
    if (Pl > 0) VIPMI_CHECK_RCCL_IN_GROUP(nc.Send(frame + (size_t)y0 * N, (size_t)Pl, ncclFloat, p, comm, st));
    if (rb > ra) VIPMI_CHECK_RCCL_IN_GROUP(nc.Recv(frame + (size_t)ra * N, (size_t)(rb - ra) * N, ncclFloat, p, comm, st));
  }
  VIPMI_CHECK_RCCL(nc.GroupEnd());
  return VIPMI_OK;
}

}  // extern "C"
