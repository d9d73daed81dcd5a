/* vipmi.h -- C ABI of libvipmi.so: MI355X (gfx950) kernels for the ADI PSF-subtraction hot path
 * of vortex-exoplanet/VIP (vip_hci.psfsub.pca / pca_annular -> svd_wrapper -> project/subtract ->
 * cube_derotate -> cube_collapse).
 *
 * The reference has no FFI for this path (it is pure Python over numpy/scipy); each
 * entry point below therefore names the reference *function* it replaces (file:line under
 * /root/reference/src/vip_hci) -- that is the interface a maintainer binds with ctypes
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 on success, a negative vipmi_status otherwise; never throws.
 *     vipmi_last_error() returns a thread-local message for the last failure.
 *   - array arguments are CALLER-OWNED DEVICE pointers (e.g. torch.Tensor.data_ptr()), C-order,
 *     contiguous, unless the name ends in _host.  Sizes are explicit int64.
 *   - a vipmi_ctx owns a device id, a stream and a growable scratch workspace; one ctx per
 *     (device, stream); a ctx is not thread-safe, distinct ctxs are independent: calls on different
 *     ctxs may be issued from different host threads at the same time and return the results of
 *     the single-threaded call bit for bit (tests/test_gpu_threads.py; the kernels of two ctxs do
 *     share compute units -- see csrc/common.h VIPMI_NO_PK32 for the one hardware interaction that
 *     this needed care for).
 *   - all work is enqueued on the ctx stream; nothing synchronises unless stated.
 */
#ifndef VIPMI_H
#define VIPMI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vipmi_ctx vipmi_ctx;

typedef enum {
  VIPMI_OK = 0,
  VIPMI_ERR_ARG = -1,      /* bad argument (maps to TypeError/ValueError on the Python side) */
  VIPMI_ERR_HIP = -2,      /* HIP runtime error */
  VIPMI_ERR_NOMEM = -3,    /* workspace allocation failed */
  VIPMI_ERR_NOCONV = -4,   /* eigensolver did not converge */
  VIPMI_ERR_UNSUPPORTED = -5
} vipmi_status;

enum { VIPMI_SCALE_TEMP_MEAN = 1, VIPMI_SCALE_TEMP_STANDARD = 2,
       VIPMI_SCALE_SPAT_MEAN = 3, VIPMI_SCALE_SPAT_STANDARD = 4 };
enum { VIPMI_COLLAPSE_MEDIAN = 0, VIPMI_COLLAPSE_MEAN = 1, VIPMI_COLLAPSE_SUM = 2,
       VIPMI_COLLAPSE_MAX = 3, VIPMI_COLLAPSE_ABSMEAN = 4, VIPMI_COLLAPSE_WMEAN = 5,
       VIPMI_COLLAPSE_TRIMMEAN = 6,
       VIPMI_COLLAPSE_STIM = 7 /* mean / population std over the frames, 0 where std == 0 (metrics/stim.py:24-44) */ };
enum { VIPMI_ROT_AUTO = 0, VIPMI_ROT_DIRECT = 1, VIPMI_ROT_FFT = 2 };

int vipmi_version(void);
const char* vipmi_last_error(void);

/* ctx: device = HIP ordinal; stream = hipStream_t (NULL = default stream). */
int vipmi_create(int device, void* stream, vipmi_ctx** out);
int vipmi_destroy(vipmi_ctx* ctx);
/* Synchronise the stream and hipFree every workspace of the context; the handle, its options, timers and gate stay
 * valid and the next call re-allocates what it needs (what a cache of contexts does to its least recently used entry). */
int vipmi_trim(vipmi_ctx* ctx);
int vipmi_set_stream(vipmi_ctx* ctx, void* stream);
int vipmi_synchronize(vipmi_ctx* ctx);
/* With option "eigh_check"=0 calls never synchronise; convergence failures are latched on the device.
 * vipmi_check_deferred synchronises the stream and returns VIPMI_ERR_NOCONV if any occurred since the last check, VIPMI_ERR_HIP
 * if a cooperating eigensolver kernel timed out waiting for a participant AND could not be recovered (more than 512 frames, or
 * option "eigh_recover"=0); recovered time-outs are not errors, vipmi_get_option(ctx, "eigh_recovered") counts them. */
int vipmi_check_deferred(vipmi_ctx* ctx);

/* Pipelining independent pca calls issued on several streams (one ctx per stream, asynchronous mode).  Contexts
 * that share a gate run the chip-filling second half of vipmi_pca_fullframe_f32 (project/subtract, derotation,
 * collapse) one call at a time, in issue order, while the latency-bound eigensolvers of the other calls run
 * beside it on a few CUs -- without the gate identical calls drift into lock step (all in the eigensolver, then
 * all in the derotation) and the chip idles.  Purely a scheduling hint: results do not depend on it. */
typedef struct vipmi_gate vipmi_gate;
int vipmi_gate_create(vipmi_gate** out);
int vipmi_gate_destroy(vipmi_gate* gate);
int vipmi_set_gate(vipmi_ctx* ctx, vipmi_gate* gate_or_null);
/* tuning knobs (key/value); see DESIGN.md.  Unknown key -> VIPMI_ERR_ARG. */
int vipmi_set_option(vipmi_ctx* ctx, const char* key, int64_t value);
int64_t vipmi_get_option(vipmi_ctx* ctx, const char* key);   /* -1 if unset */
/* With option "timing"=1 every stage / kernel records hipEvent pairs on the ctx stream.
 * vipmi_stage_ms: total elapsed ms of all intervals of the named stage since the last
 * vipmi_reset_timers (synchronises; <0 if none); vipmi_stage_count: number of intervals.
 * Stages: "gram","eigh","project","derotate","collapse","scale"; single kernels:
 * "k_gram","k_rowspace","k_subtract","k_rot_s1","k_rot_s2","k_rot_s3","k_median".
 * "timing"=3: no events; vipmi_stage_ms returns the HOST wall time spent inside each stage (enqueue cost; includes
 * waiting for a staging slot when the host runs ahead of the GPU) -- tools/time_small_calls.py. */
float vipmi_stage_ms(vipmi_ctx* ctx, const char* stage);
int vipmi_stage_count(vipmi_ctx* ctx, const char* stage);
int vipmi_reset_timers(vipmi_ctx* ctx);

/* ---- prepare_matrix pieces: var/shapes.py:740-781 (matrix_scaling), :38-113 (mask_circle) ---- */
/* out[n,P] = sklearn-style scale of in[n,P]; mode = VIPMI_SCALE_*; in == out allowed. */
int vipmi_scale_f32(vipmi_ctx* ctx, const float* in, float* out, int64_t n, int64_t P, int mode);
/* out[n,P] = in[n,P] with pixels whose mask byte != 0 set to fill (mask computed by the host with the
 * reference's float64 disk rule so membership is bit-exact). */
int vipmi_apply_mask_f32(vipmi_ctx* ctx, const float* in, float* out, int64_t n, int64_t P,
                         const uint8_t* mask, float fill);

/* `batch` Gram matrices of equal shape in one launch: M is [batch][n][P], G is [batch][n][n] (float64).  Used for
 * the first pass of ADI+mSDI (pca_fullfr.py:1482-1520: one spectral PCA per multispectral frame). */
int vipmi_gram_batched_f32(vipmi_ctx* ctx, const float* M, int64_t batch, int64_t n, int64_t P, double* G);

/* ---- svd_wrapper mode 'eigen'/'lapack' arithmetic: psfsub/svd.py:447-475 ---- */
/* G[n,n] (float64, symmetric) = M[n,P] * M^T, M row-major with leading dimension ld (floats). */
int vipmi_gram_f32(vipmi_ctx* ctx, const float* M, int64_t n, int64_t P, int64_t ld, double* G);
/* Cross product C[na,nb] (float64) = A[na,P] * B[nb,P]^T (used by RDI / cube_sig projections). */
int vipmi_cross_gram_f32(vipmi_ctx* ctx, const float* A, int64_t na, const float* B, int64_t nb,
                         int64_t P, int64_t ld, double* C);
/* Batched symmetric eigendecomposition (float64, one-sided block Jacobi).  G: batch x n x n,
 * destroyed.  evals: batch x n descending.  evecs: batch x n x n, row i = eigenvector of evals[i]
 * (unit norm, sign: largest-|component| positive).  Rank-deficient input: the rows of eigenvalues that are numerically null
 * (below ~1e-13 of the largest column norm) are unit vectors without meaning -- the PCA callers drop every component below 1e-12
 * of the largest eigenvalue (the reference divides by its singular value, psfsub/svd.py:447-475). */
int vipmi_eigh_f64(vipmi_ctx* ctx, double* G, int64_t batch, int64_t n, double* evals, double* evecs);

/* Leading k eigenpairs only (k <= 64, n <= 512: Householder tridiagonalisation + multisection + inverse iteration,
 * one workgroup per problem; other sizes fall back to vipmi_eigh_f64).  Same layout as vipmi_eigh_f64 for the first
 * k rows of evals / evecs; the other entries are unspecified.  nact (device int32[batch], may be NULL): active
 * leading size of each zero-padded problem.  Replaces get_eigenvectors(ncomp, ...) (psfsub/svd.py:623-702) and the
 * truncated decompositions of svd_wrapper (svd.py:342-620). */
int vipmi_eigh_topk_f64(vipmi_ctx* ctx, double* G, int64_t batch, int64_t n, int64_t k, const int32_t* nact,
                        double* evals, double* evecs);
/* Verified fast path alone (Chebyshev-filtered block subspace iteration on the float64 matrix cores, csrc/eigh_chfsi.hip;
 * replaces the truncated decompositions of psfsub/svd.py:447-491,705-808 where the spectrum allows): leading k pairs of ONE
 * symmetric positive semi-definite G[n,n] (not modified).  *converged = 1: evals[0..k) descending, evecs[k,n] rows, every pair
 * with ||G q - theta q|| <= 1e-13 theta_1; *converged = 0: nothing written -- no usable gap behind the k-th eigenvalue, or
 * sizes outside 256 <= n <= 16384, k + max(12, k/4) <= 64, 4k <= n.  vipmi_eigh_topk_f64 tries it by itself for one matrix of
 * 700 .. 6144 rows -- from 800 / 1000 for blocks of 48 / 64 vectors -- (option "eigh_fast", default 1) and falls back on the exact tridiagonal path. */
int vipmi_eigh_topk_fast_f64(vipmi_ctx* ctx, const double* G, int64_t n, int64_t k, double* evals, double* evecs,
                             int* converged);

/* All n eigenvalues (descending) and the leading k eigenvectors: what SVDecomposer.get_cevr (psfsub/svd.py:216-339) and
 * svd_wrapper(..., full_output=True) in the 'eigen' modes (svd.py:454-462) need.  Same layout as vipmi_eigh_f64. */
int vipmi_eigh_spectrum_f64(vipmi_ctx* ctx, double* G, int64_t batch, int64_t n, int64_t k, double* evals,
                            double* evecs);

/* ---- _project_subtract: psfsub/pca_fullfr.py:1727-1731 ---- */
/* B[k,P] = W[k,n] (float32) * M[n,P];  row c optionally scaled by rowscale[c] (may be NULL). */
int vipmi_rowspace_gemm_f32(vipmi_ctx* ctx, const float* W, const float* M, int64_t k, int64_t n,
                            int64_t P, const float* rowscale, float* B);
/* R[n,P] = M[n,P] - C[n,k]*B[k,P];  recon (may be NULL) receives C*B. */
int vipmi_subtract_gemm_f32(vipmi_ctx* ctx, const float* M, const float* C, const float* B,
                            int64_t n, int64_t k, int64_t P, float* R, float* recon);

/* out = a*x + b*y (y may be NULL: out = a*x), float32, `total` elements: the elementwise glue of the reference's
 * numpy expressions (reconstructed = matrix - residuals, pca_fullfr.py:1731; STIM normalisation, metrics/stim.py:118). */
int vipmi_lincomb_f32(vipmi_ctx* ctx, const float* x, const float* y, float a, float b, int64_t total, float* out);

/* ---- FFT zoom of the spectral channels (ADI+mSDI): scale_fft / frame_rescaling(imlib='vip-fft') /
 * cube_rescaling_wavelengths, preproc/rescaling.py:1114-1217, 636-672, 427-475 ----
 * The zoom is a separable linear map Y = Re(E X E^T); E (dout x din, complex, one per scale factor, reflect padding and
 * crops folded in) is built on the host.  out[b] (dout x dout) = Er[c] X[b] Er[c]^T - Ei[c] X[b] Ei[c]^T with
 * c = chan[b]; X[b] is din x din.  Er / Ei: [nchan][dout][ldk] row-major with ldk = din rounded up to 4 (zero
 * padded); X: [nb][din][din]; work: float32 scratch of nb * 2 * dout * ldk (device). */
int vipmi_zoom_frames_f32(vipmi_ctx* ctx, const float* X, int64_t nb, int64_t din, const float* Er, const float* Ei,
                          const int32_t* chan, int64_t dout, int64_t ldk, float* work, float* out);

/* ---- cube_derotate / frame_rotate(imlib='vip-fft'): preproc/derotation.py:51-328,331-399,542-640 ----
 * out[n,N,N] = frames of in[n,N,N] rotated by -angles_host[i] degrees with the reference's 3-shear
 * FFT rotation (1.5x then 4x zero padding, rot90 pre-step, complex field carried between shears).
 * NaN input pixels are treated as 0 and restored as NaN in the output when mask_nan != 0;
 * when mask_zero != 0 pixels equal to 0 in the input are restored to 0 (mask_val=0 semantics).
 * method = VIPMI_ROT_*. */
int vipmi_derotate_f32(vipmi_ctx* ctx, const float* in, const double* angles_host, int64_t n,
                       int64_t N, float* out, int mask_nan, int mask_zero, int method);
/* The same with frame_rotate's `mask_val` itself (derotation.py:133-140,324-326): NaN -> the mask_nan behaviour above;
 * any other value v -> NaN input pixels are rotated as 0 (and NOT restored), pixels equal to v take part in the
 * rotation with their value and are reset to v in the output. */
int vipmi_derotate_maskval_f32(vipmi_ctx* ctx, const float* in, const double* angles_host, int64_t n,
                               int64_t N, float* out, float mask_val, int method);

/* ---- cube_derotate / frame_rotate(imlib='opencv'): preproc/derotation.py:279-305 ----
 * out[n,N,N] = frames of in[n,N,N] rotated by -angles_host[i] degrees about (cx, cy) with OpenCV's
 * getRotationMatrix2D + warpAffine arithmetic (float32, 1/32-pixel phases, border = VIPMI_BORDER_*; NaN -> 0).
 * interp = VIPMI_INTERP_* ('nearneig' | 'bilinear' | 'bicubic' | 'lanczos4').  The fast, lower-fidelity
 * alternative to vipmi_derotate_f32 (README.rst:183); cv2 is absent from the build image, so parity with
 * cv2 itself is unpinned (see oracle/ref_cpu.py warp_rotate). */
enum { VIPMI_INTERP_NEAREST = 0, VIPMI_INTERP_BILINEAR = 1, VIPMI_INTERP_BICUBIC = 2, VIPMI_INTERP_LANCZOS4 = 3 };
/* border_mode 'constant' | 'edge' | 'symmetric' | 'reflect' | 'wrap' = cv2.BORDER_CONSTANT (0) | REPLICATE | REFLECT |
 * REFLECT_101 | WRAP (derotation.py:294-305) */
enum { VIPMI_BORDER_CONSTANT = 0, VIPMI_BORDER_REPLICATE = 1, VIPMI_BORDER_REFLECT = 2, VIPMI_BORDER_REFLECT101 = 3,
       VIPMI_BORDER_WRAP = 4 };
int vipmi_rotate_interp_f32(vipmi_ctx* ctx, const float* in, const double* angles_host, int64_t n, int64_t N,
                            double cx, double cy, int interp, int border, float* out);

/* ---- cube_collapse: preproc/subsampling.py:30-116 ---- */
/* out[P] = collapse over the n frames of cube[n,P]; NaN-aware (nanmedian/nanmean/...).
 * w (device, n floats) only for WMEAN; trim_n only for TRIMMEAN. */
int vipmi_collapse_f32(vipmi_ctx* ctx, const float* cube, int64_t n, int64_t P, int mode,
                       const float* w, int64_t trim_n, float* out);
/* the same for `batch` contiguous cubes [batch][n][P] -> out[batch][P] in one launch: the per-channel collapses of a
 * 4-D cube (pca_fullfr.py:544-658) and the per-frame channel collapses of ADI+mSDI (pca_fullfr.py:1339-1344,1519) */
int vipmi_collapse_batched_f32(vipmi_ctx* ctx, const float* cubes, int64_t batch, int64_t n, int64_t P, int mode,
                               const float* w, int64_t trim_n, float* out);

/* ---- median_sub(mode='annular') core: psfsub/medsub.py:602-641 ----
 * out[j,p] = A[j,p] - nanmedian over the frames lib_idx[j, 0 .. lib_len[j]) of A[.,p]  (A: n x npx annulus matrix,
 * lib_idx: [n][max_lib] device int32, max_lib <= 32: the `nframes` closest frames beyond the PA threshold). */
int vipmi_subset_median_sub_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* lib_idx,
                                const int32_t* lib_len, int64_t max_lib, float* out);

/* ---- pca_annular core: psfsub/pca_local.py:710-787,830-909 ----
 * For one annulus segment matrix A[n,npx] (already gathered + scaled) and per-frame library index
 * lists (lib_idx[n*max_lib], lib_len[n], device int32), computes residuals[n,npx] =
 * A[j] - proj_{top-k PCs of A[lib_j]}(A[j]) through the sub-Gram identity. */
int vipmi_annular_residuals_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx,
                                const int32_t* lib_idx, const int32_t* lib_len, int64_t max_lib,
                                int64_t ncomp, float* residuals);
/* Same with a LIST of truncation ranks (pca_local.py:665-668,892-902: one decomposition with max(ncomp), one
 * residual matrix per V[:k]): ncomps_host is a HOST array of nk ranks, residuals is [nk][n][npx] (device). */
int vipmi_annular_residuals_multi_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx,
                                      const int32_t* lib_idx, const int32_t* lib_len, int64_t max_lib,
                                      const int32_t* ncomps_host, int64_t nk, float* residuals);
/* The same computation in stages, so that a caller can solve the libraries of ALL segments of a frame set in ONE
 * vipmi_eigh_topk_f64 call (do_pca_patch of every segment, pca_local.py:830-909; 3200 eigenproblems at BASELINE C3):
 * subgrams: G[n,n] = A A^T (float64) and the zero-padded library sub-Gram matrices H[n][m][m], m >= max_lib;
 * apply: residuals[nk][n][npx] from the leading eigenpairs evals[n][m], evecs[n][m][m] (rows = vectors, descending). */
int vipmi_annular_subgrams_f64(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* lib_idx,
                               const int32_t* lib_len, int64_t max_lib, int64_t m, double* G, double* H);
/* eigh (round 5): the eigensolve of ALL libraries of nseg segments (n frames each) in one call WITHOUT the sub-Gram matrices of
 * `subgrams` -- the solver gathers problem p = seg n + j as G[seg][idx[p][a]][idx[p][b]], a, b < len[p], from the segments' Gram
 * matrices G[nseg][n][n] (vipmi_gram_f32 of every segment matrix).  lib_idx: [nseg n][m] int32 (rows padded to m >= every
 * library), lib_len: [nseg n]; work: [nseg n][m][m] float64 workspace; evals [nseg n][m], evecs [nseg n][m][m] as
 * vipmi_eigh_topk_f64 returns them (leading k).  do_pca_patch, pca_local.py:830-909. */
int vipmi_annular_eigh_f64(vipmi_ctx* ctx, const double* G, int64_t nseg, int64_t n, const int32_t* lib_idx,
                           const int32_t* lib_len, int64_t m, int64_t k, double* work, double* evals, double* evecs);
/* The fronts of ALL segments of an annular PCA in a handful of launches (round 6; _pca_adi_rdi's loop over annuli and segments,
 * pca_local.py:710-787, with do_pca_patch :830-909 for every frame of every segment):
 * gram_all: A_all[n][Ptot] = the segment matrices side by side -- column p is pixel pix_all[p] of the cube (flat index; -1 = zero
 *   column), every segment padded to a whole number of K-slices of klen columns (klen a multiple of 64 in 256..4096, Ptot a multiple
 *   of klen) -- and their Gram matrices G_all[nseg][n][n] (float64) in ONE ragged product on the int8 matrix cores (the exact-integer
 *   scheme of vipmi_gram_f32, 3e-12 of max|G|); seg_slice: device int32[nseg + 1], the segments' first slices (seg_slice[nseg] =
 *   Ptot / klen).  cube NULL: A_all already holds the matrix (a caller that scaled it first, matrix_scaling of every segment).
 * apply_all: after vipmi_annular_eigh_f64 on G_all -- the coefficient matrices of all segments and ONE product
 *   residuals = (I - C_seg) A_seg written straight into cube_out[n][P] through pix_out[Ptot] (as pix_all, with -1 also for the pixels
 *   a LATER segment owns: the reference applies the segments in order, pca_local.py:786-787); tile_seg: device int32[Ptot / 128], the
 *   segment of every 128-column tile (-1: padding only); kseg: device int32[nseg], min(ncomp, pixels) of every segment, <= kmax;
 *   mu32: NULL, or float32[Ptot] for the float64 front below (the rank-one term rho mu^T is added on the way out). */
int vipmi_annular_gram_all_f32(vipmi_ctx* ctx, const float* cube, int64_t n, int64_t P, const int32_t* pix_all, int64_t Ptot,
                               int64_t klen, const int32_t* seg_slice, int64_t nseg, float* A_all, double* G_all);
int vipmi_annular_apply_all_f32(vipmi_ctx* ctx, const float* A_all, int64_t n, int64_t Ptot, const int32_t* tile_seg,
                                const int32_t* pix_out, int64_t nseg, const int32_t* lib_idx, const int32_t* lib_len, int64_t m,
                                const double* G_all, const double* evals, const double* evecs, const int32_t* kseg, int64_t kmax,
                                int64_t P, float* cube_out, const float* mu32);
/* gram_all for a FLOAT64 cube (the reference's do_pca_patch keeps the caller's dtype, pca_local.py:830-909): the gather and the
 * centring of vipmi_center_f64 in one pass -- D_all[n][Ptot] = float32((cube[:, pix_all] - 1 mu^T) / sd), mode 0 / 1: centre, 2:
 * 'temp-standard'; mu[Ptot] float64, mu32[Ptot] = float32(mu) --, the ragged Gram product of D_all, and for mode 0 (no scaling) the
 * float64 offset terms of vipmi_gram_offset_f64 for every segment: G_all[seg] is the Gram matrix of D + 1 mu^T.  apply_all then
 * takes A_all = D_all and, for mode 0, mu32 (NULL otherwise): residuals = (I - C) D + rho mu^T, rho = (I - C) 1 in float64. */
int vipmi_annular_gram_all_f64(vipmi_ctx* ctx, const double* cube, int64_t n, int64_t P, const int32_t* pix_all, int64_t Ptot,
                               int64_t klen, const int32_t* seg_slice, int64_t nseg, int mode, float* D_all, double* mu, float* mu32,
                               double* G_all);
/* float64 cubes (round 5): the per-pixel temporal mean -- what float32 cannot hold beside the signal in a cube of detector counts --
 * is carried in float64 (csrc/pca_f64.hip; the reference keeps the caller's dtype through svd_wrapper / do_pca_patch):
 * center: D[n][P] = float32((M - 1 mu^T) / sd), mu[P] float64, mu32[P] = float32(mu) (optional); mode 0 / 1: centre ('temp-mean'),
 *   2: 'temp-standard';  gram_offset: G (= D D^T from vipmi_gram_f32) += 1 (D mu)^T + (D mu) 1^T + |mu|^2 1 1^T, the Gram matrix of
 *   D + 1 mu^T (scaling None);  annular_apply_mu: vipmi_annular_apply_f32 on D with residuals += rho mu32^T, rho = (I - C) 1 (mu32 NULL:
 *   plain apply). */
int vipmi_center_f64(vipmi_ctx* ctx, const double* M, int64_t n, int64_t P, int mode, float* D, double* mu, float* mu32);
int vipmi_gram_offset_f64(vipmi_ctx* ctx, const float* D, const double* mu, int64_t n, int64_t P, double* G);
/* The SPATIAL scalings (matrix_scaling axis=1, var/shapes.py:740-781) of a float64 matrix M[n][P] whose first Preal columns are
 * samples (the rest zero padding): the scaled matrix diag(u) (M - m 1^T) -- m the rows' means, u their inverse standard deviations
 * (with_std = 0, 'spat-mean': ones) -- is D + u mu^T with D[n][P] = float32 of diag(u) [(M - 1 mu0^T) - (m - mean(m) 1) 1^T] formed
 * in float64, mu[P] = mu0 - mean(mu0) (mu0 = the columns' means; zero in the padding), mu32 = float32(mu), u[n] float64 (device).
 * gram_offset_u: G (= D D^T) += u (D mu)^T + (D mu) u^T + |mu|^2 u u^T; annular_apply_mu_u: residuals = (I - C) D + rho mu32^T with
 * rho = (I - C) u.  u NULL in either: ones, i.e. vipmi_gram_offset_f64 / vipmi_annular_apply_mu_f32. */
int vipmi_spat_center_f64(vipmi_ctx* ctx, const double* M, int64_t n, int64_t P, int64_t Preal, int with_std, float* D, double* mu,
                          float* mu32, double* u);
int vipmi_gram_offset_u_f64(vipmi_ctx* ctx, const float* D, const double* mu, const double* u, int64_t n, int64_t P, double* G);
int vipmi_annular_apply_mu_u_f32(vipmi_ctx* ctx, const float* D, int64_t n, int64_t npx, const int32_t* lib_idx,
                                 const int32_t* lib_len, int64_t max_lib, int64_t m, const double* G, const double* evals,
                                 const double* evecs, const int32_t* ncomps_host, int64_t nk, const float* mu32, const double* u,
                                 float* residuals);
int vipmi_annular_apply_mu_f32(vipmi_ctx* ctx, const float* D, int64_t n, int64_t npx, const int32_t* lib_idx,
                               const int32_t* lib_len, int64_t max_lib, int64_t m, const double* G, const double* evals,
                               const double* evecs, const int32_t* ncomps_host, int64_t nk, const float* mu32,
                               float* residuals);
int vipmi_annular_apply_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* lib_idx,
                            const int32_t* lib_len, int64_t max_lib, int64_t m, const double* G, const double* evals,
                            const double* evecs, const int32_t* ncomps_host, int64_t nk, float* residuals);
/* gather / scatter of annulus pixels: A[n,npx] = cube[n, pix[j]] and back. */
int vipmi_gather_f32(vipmi_ctx* ctx, const float* cube, int64_t n, int64_t P, const int32_t* pix,
                     int64_t npx, float* A);
int vipmi_scatter_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t P, const int32_t* pix,
                      int64_t npx, float* cube);

/* ---- svd_wrapper + _project_subtract in one call: psfsub/svd.py:342-620, pca_fullfr.py:1717-1731 ----
 * residuals[n,P] = M - proj(M onto the top-k PCs of ref[nref,P]) ; ref == M (same pointer) is ADI,
 * otherwise RDI.  Optional outputs (may be NULL): recon[n,P], pcs[k,P] (orthonormal rows, the
 * reference's V), evals_out[nref] (float64 eigenvalues of ref ref^T = squared singular values). */
int vipmi_pca_project_f32(vipmi_ctx* ctx, const float* M, int64_t n, const float* ref, int64_t nref,
                          int64_t P, int64_t k, float* residuals, float* recon, float* pcs,
                          double* evals_out);

/* R[b] = M[b] - E[b]^T (E[b] M[b]) for a contiguous batch of equally shaped problems: M, R [nb][n][P] and
 * E [nb][k][n] (rows = leading eigenvectors of M[b] M[b]^T from vipmi_eigh_topk_f64, in float32).  The projection of
 * pca_fullfr.py:1727-1731 for every multispectral frame of ADI+mSDI (:1482-1520) / every channel of a 4-D cube
 * (:544-658) in two launches. */
int vipmi_project_batched_f32(vipmi_ctx* ctx, const float* M, const float* E, int64_t nb, int64_t n, int64_t k,
                              int64_t P, float* R);

/* ---- fused full-frame ADI path: psfsub/pca_fullfr.py:801-1007 (3-D, int ncomp, no cube_ref) ----
 * cube[n,N,N] float32 -> frame[N,N].  Optional outputs may be NULL: pcs[k,N,N], recon[n,N,N],
 * residuals[n,N,N], residuals_der[n,N,N].  scaling: 0 or VIPMI_SCALE_*.  mask: N*N bytes or NULL. */
int vipmi_pca_fullframe_f32(vipmi_ctx* ctx, const float* cube, const double* angles_host, int64_t n,
                            int64_t N, int64_t ncomp, int scaling, const uint8_t* mask,
                            int collapse_mode, float* frame, float* pcs, float* recon,
                            float* residuals, float* residuals_der);
/* The same for a FLOAT64 cube (device pointer); outputs as vipmi_pca_fullframe_f32 (float32; pcs [k][N][N], recon / residuals / residuals_der [n][N][N], each optional).  The reference keeps the
 * caller's dtype through prepare_matrix / svd_wrapper (psfsub/pca_fullfr.py:1552-1737, psfsub/svd.py:342-620); here the per-pixel
 * temporal mean -- the part of a cube of detector counts that float32 cannot hold beside the signal -- is carried in float64:
 * D = float32(cube - 1 mu^T) goes through the float32 kernels, the decomposition is that of D + 1 mu^T (Gram corrected in float64),
 * residual = [D - E^T (E D)] + (1 - E^T E 1) mu^T.  scaling: 0 (None) or any VIPMI_SCALE_*; the spatial scalings (round 6:
 * matrix_scaling with axis=1, var/shapes.py:740-781) have the same shape with the frames' inverse standard deviations u in the
 * place of 1:  diag(u) (M - m 1^T) = D + u (mu - mean(mu))^T,  D = float32 of diag(u) [(M - 1 mu^T) - (m - mean(m)) 1^T] formed in float64. */
int vipmi_pca_fullframe_f64(vipmi_ctx* ctx, const double* cube, const double* angles_host, int64_t n, int64_t N, int64_t ncomp,
                            int scaling, const uint8_t* mask, int collapse_mode, float* frame, float* pcs, float* recon,
                            float* residuals, float* residuals_der);

/* The fused path for a cube still in HOST memory (the reference's callers pass numpy arrays: psfsub/pca_fullfr.py:137,
 * metrics/contrcurve.py:768-790).  host_cube[n][N][N] float32 (pageable or pinned) is copied into the device buffer `cube`
 * (kept by the caller: n*N*N floats) in blocks of 64 frames on a copy stream of the context, and behind every block the Gram
 * matrix of the blocks that have arrived is advanced on the context's stream -- where vipmi_gram_f32 would take the int8 path by
 * itself; otherwise: plain copy, then vipmi_pca_fullframe_f32.  mask (N*N bytes on the device, or NULL) as in
 * vipmi_pca_fullframe_f32: every block is masked as it arrives.  No scaling (the temporal statistics need every frame: upload,
 * then call vipmi_pca_fullframe_f32).  Results bit-identical to vipmi_pca_fullframe_f32 on the uploaded cube.  A pageable host
 * array has been read completely when the call returns; a pinned one once the context's stream has passed the call. */
int vipmi_pca_fullframe_hostin_f32(vipmi_ctx* ctx, const float* host_cube, float* cube, const double* angles_host, int64_t n,
                                   int64_t N, int64_t ncomp, const uint8_t* mask, int collapse_mode, float* frame, float* pcs,
                                   float* recon, float* residuals, float* residuals_der);

/* ---- 4-D (IFS) cube without scale_list: psfsub/pca_fullfr.py:544-658 ----
 * cube4[nch,n,N,N] float32: one full-frame ADI PCA per spectral channel (same integer ncomp, no reference cube), then
 * the spectral collapse (collapse_ifs) of the nch per-channel frames -> frame[N,N].  ifs_frames[nch,N,N] (the
 * reference's ifs_adi_frames) is optional (NULL).  The per-channel stages are batched: one Gram, one eigensolver, two
 * projection, one derotation and one collapse launch for all channels.  n <= 512 frames and <= 64 PCs per channel
 * (else VIPMI_ERR_ARG: loop vipmi_pca_fullframe_f32 over the channels). */
int vipmi_pca_4d_f32(vipmi_ctx* ctx, const float* cube4, const double* angles_host, int64_t nch, int64_t n, int64_t N,
                     int64_t ncomp, int scaling, const uint8_t* mask, int collapse_mode, int collapse_ifs_mode,
                     float* frame, float* ifs_frames);

/* ---- one cube over several GPUs ("C2/C5 single cube"): psfsub/pca_fullfr.py:801-1007 with the pixels
 * sharded for the decomposition and the frames sharded for the derotation.  One process per GPU; RCCL is resolved at
 * run time (dlopen; `path` NULL = the copy already in the process, else the loader's search path).
 *   vipmi_rccl_unique_id:    128-byte ncclUniqueId, made on one rank and handed to the others by the host program
 *   vipmi_rccl_comm_create:  ncclCommInitRank on ctx's device; *comm is an ncclComm_t (a communicator made elsewhere
 *                            with the SAME RCCL library may be passed to the call below instead)
 *   vipmi_pca_fullframe_sharded_f32: every rank passes its row slab of the cube, slab[n][y1-y0][N] with
 *       [y0, y1) = rows of rank r of `world` contiguous near-equal blocks (block r starts at r*(N/world) + min(r, N%world)),
 *       and the whole angle vector; frame[N,N] (device) receives the final frame on EVERY rank.  Collectives on
 *       ctx's stream: one all-reduce of n*n float64, two all-to-alls of the residual cube, one exchange of N*N floats.
 *       Unlike vipmi_pca_fullframe_f32 this entry takes no `scaling` and no `mask` (scale / mask the slab first with
 *       vipmi_scale_f32 / vipmi_apply_mask_f32 -- the temporal modes are per pixel and shard with the rows; the
 *       spatial modes need whole frames) and no weighted collapse; at most 6144 frames, checked before any collective. */
int vipmi_rccl_load(const char* path);
int vipmi_rccl_unique_id(void* id128);
int vipmi_rccl_comm_create(vipmi_ctx* ctx, const void* id128, int rank, int world, void** comm);
int vipmi_rccl_comm_destroy(void* comm);
int vipmi_pca_fullframe_sharded_f32(vipmi_ctx* ctx, void* comm, int rank, int world, const float* slab,
                                    const double* angles_host, int64_t n, int64_t N, int64_t ncomp, int collapse_mode,
                                    float* frame);

#ifdef __cplusplus
}
#endif
#endif /* VIPMI_H */
