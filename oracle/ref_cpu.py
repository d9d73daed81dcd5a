"""CPU oracle for the ADI PSF-subtraction hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-numpy restatement of the algorithm that ``vip_hci.psfsub.pca`` /
``pca_annular`` execute (reference = vortex-exoplanet/VIP 2.0.1, mounted read-only at
``/root/reference`` in the build container).  Every function cites the reference
``file:line`` it restates.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the checker /
baseline -- the product (``vip_amd``) never falls back to it.

Pinning: ``oracle/check_vs_reference.py`` compares every function below with the imported
reference (via ``oracle/_shim.py``) and ``oracle/gen_golden.py`` freezes reference outputs
into ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` re-checks this file against
those fixtures everywhere (no reference needed).

The restatement is written from the algorithm's definition, not
transliterated: e.g. the rotation is expressed as per-line circular sinc shifts instead
of the reference's fftshift/tile formulation (they agree to ~1e-14, see the check script).
"""
import numpy as np

# --------------------------------------------------------------------------------------
# geometry helpers
# --------------------------------------------------------------------------------------


def frame_center(shape_or_array):
    """(cy, cx) = (ny//2, nx//2) for both parities.  Ref: var/coords.py:61-100."""
    shp = shape_or_array.shape[-2:] if hasattr(shape_or_array, "shape") else tuple(shape_or_array)[-2:]
    return int(shp[0] // 2), int(shp[1] // 2)


def check_pa_vector(angle_list):
    """Ref: preproc/parangles.py:405-458 (unit='deg')."""
    a = np.array(angle_list, dtype=float, copy=True)
    a[a < 0] += 360.0
    if a.size > 1 and np.any(np.abs(np.diff(a)) > 180):
        a[a < 180] += 360.0
    return a


def disk_indices(center, radius, shape):
    """Pixels with ((r-cy)/R)^2 + ((c-cx)/R)^2 < 1 (scikit-image ``draw.disk`` rule),
    row-major order.  Third-party dependency of the reference (scikit-image 0.25.2,
    not vendored); used at var/shapes.py:88."""
    cy, cx = center
    if shape is None:                     # unbounded (metrics use disk((y, x), fwhm/2) without a shape)
        rr, cc = np.mgrid[int(np.ceil(cy - radius)):int(np.floor(cy + radius)) + 1,
                          int(np.ceil(cx - radius)):int(np.floor(cx + radius)) + 1]
    else:
        rr, cc = np.mgrid[:shape[0], :shape[1]]
    m = ((rr - cy) / radius) ** 2 + ((cc - cx) / radius) ** 2 < 1
    return rr[m], cc[m]


def mask_circle(array, radius, fillwith=0):
    """mode='in' only.  Ref: var/shapes.py:38-113.  Note the reference indexes
    ``[:, ind[1], ind[0]]`` for 3d/4d (transposed; symmetric for a centred disk)."""
    if radius == 0:
        return True * array
    cy, cx = frame_center(array)
    shape = array.shape[-2:]
    ind = disk_indices((cy, cx), radius, shape)
    out = array.copy()
    if array.ndim == 2:
        out[ind] = fillwith
    elif array.ndim == 3:
        out[:, ind[1], ind[0]] = fillwith
    else:
        out[:, :, ind[1], ind[0]] = fillwith
    return out


def matrix_scaling(matrix, scaling):
    """sklearn.preprocessing.scale (1.7.2) arithmetic restated.
    Ref: var/shapes.py:740-781; sklearn/preprocessing/_data.py::scale."""
    if scaling is None:
        return matrix
    table = {"temp-mean": (0, False), "spat-mean": (1, False),
             "temp-standard": (0, True), "spat-standard": (1, True)}
    if scaling not in table:
        raise ValueError("Scaling mode not recognized")
    axis, with_std = table[scaling]
    X = np.array(matrix, copy=True)
    if X.dtype not in (np.float32, np.float64):
        X = X.astype(np.float64)
    mean_ = np.nanmean(X, axis)
    if with_std:
        scale_ = np.nanstd(X, axis)
    Xr = np.rollaxis(X, axis)          # view: broadcasting axis first
    Xr -= mean_
    mean_1 = np.nanmean(Xr, axis=0)
    if not np.allclose(mean_1, 0):
        Xr -= mean_1
    if with_std:
        scale_ = np.array(scale_, copy=True)
        scale_[scale_ < 10 * np.finfo(scale_.dtype).eps] = 1.0
        Xr /= scale_
        mean_2 = np.nanmean(Xr, axis=0)
        if not np.allclose(mean_2, 0):
            Xr -= mean_2
    return X


def prepare_matrix(cube, scaling=None, mask_center_px=None):
    """mode='fullfr'.  Ref: var/shapes.py:857-873."""
    arr = cube
    if mask_center_px:
        arr = mask_circle(arr, mask_center_px)
    m = np.reshape(arr, (arr.shape[0], -1))
    return matrix_scaling(m, scaling)


# --------------------------------------------------------------------------------------
# SVD / projection
# --------------------------------------------------------------------------------------


def randomized_svd(M, k, n_iter=2, n_oversamples=10, seed=0):
    """Halko et al. randomized SVD as sklearn.utils.extmath.randomized_svd runs it for the
    reference's call (svd.py:487-491): transpose='auto' (work on M.T when rows < cols),
    power_iteration_normalizer='auto' -> 'none' for n_iter <= 2.  The reference is
    unseeded; this restatement takes a seed (documented deviation)."""
    rng = np.random.RandomState(seed)
    A = M.T if M.shape[0] < M.shape[1] else M
    transposed = A is not M
    Q = rng.normal(size=(A.shape[1], k + n_oversamples)).astype(A.dtype)
    for _ in range(n_iter):
        Q = A @ Q
        Q = A.T @ Q
    Q, _ = np.linalg.qr(A @ Q, mode="reduced")
    B = Q.T @ A
    Uh, s, Vt = np.linalg.svd(B, full_matrices=False)
    U = Q @ Uh
    if transposed:
        return Vt[:k].T, s[:k], U[:, :k].T
    return U[:, :k], s[:k], Vt[:k]


def svd_wrapper(matrix, mode, ncomp, full_output=False, seed=0, left_eigv=False):
    """Returns V (k x P, rows = PCs) or (U, S, V); ``left_eigv`` (mode 'lapack'): the temporal modes, (n x k)
    (svd.py:607-613).  Ref: psfsub/svd.py:342-620."""
    if matrix.ndim != 2:
        raise TypeError("Input matrix is not a 2d array")
    if ncomp > min(matrix.shape):
        raise RuntimeError("{} PCs cannot be obtained from a matrix with size [{},{}]."
                           .format(ncomp, matrix.shape[0], matrix.shape[1]))
    if left_eigv and mode != "lapack":
        raise NotImplementedError("left_eigv: mode 'lapack' only in this restatement")
    if mode == "lapack":
        # svd of M.T (P x n): left vectors of M.T are the PCs (svd.py:466-475,598,615)
        Ul, S, Vl = np.linalg.svd(matrix.T, full_matrices=False)
        V = Ul[:, :ncomp].T
        if full_output:
            return Vl[:ncomp].T, S[:ncomp], V
        if left_eigv:
            return Vl[:ncomp].T
        return V
    if mode == "eigen":
        # svd.py:447-464
        C = matrix @ matrix.T
        e, EV = np.linalg.eigh(C)
        pc = EV.T @ matrix
        S = np.sqrt(np.abs(e))[::-1]
        V = pc[::-1] / S[:, None]
        V = V[:ncomp]
        if full_output:
            U = (EV / np.sqrt(np.abs(e)))[:ncomp]      # reference quirk kept (svd.py:460-462)
            return U, S, V
        return V
    if mode == "randsvd":
        U, S, V = randomized_svd(matrix, ncomp, seed=seed)
        if full_output:
            return U, S, V
        return V
    raise ValueError("The SVD `mode` is not recognized")


def cevr_to_ncomp(cube, cevr, scaling=None, svd_mode="lapack"):
    """float ncomp -> int k.  Ref: psfsub/svd.py:182-185,204-207,253-257,335 (note: the
    reference's SVDecomposer ignores mask_center_px here)."""
    m = prepare_matrix(cube, scaling, None)
    _, S, _ = svd_wrapper(m, svd_mode, min(m.shape), full_output=True)
    ev = S ** 2 / (S.shape[0] - 1)
    c = np.cumsum(ev / np.sum(ev))
    return int(np.searchsorted(c, cevr) + 1)


def project_subtract(cube, ncomp, scaling=None, mask_center_px=None, svd_mode="lapack",
                     cube_ref=None, full_output=False, seed=0, cube_sig=None, left_eigv=False):
    """Whole-matrix branch of ``_project_subtract``.  Ref: psfsub/pca_fullfr.py:1649-1737.
    ``cube_sig`` (estimated signal, :1652-1662): the PCs are learnt from and the projection is taken of the
    "empty" matrix ``matrix - reshape(cube_sig)`` (cube_sig is neither masked nor scaled), but the model is
    subtracted from ``matrix`` itself (:1717-1731)."""
    n, y, x = cube.shape
    if isinstance(ncomp, (float, np.floating)):
        if not 1 > ncomp > 0:
            raise ValueError("if `ncomp` is float, it must lie in the interval (0,1]")
        ncomp = cevr_to_ncomp(cube, ncomp, scaling, svd_mode)
    matrix = prepare_matrix(cube, scaling, mask_center_px)
    matrix_emp = matrix if cube_sig is None else matrix - np.reshape(cube_sig, (cube_sig.shape[0], -1))
    ref_lib = matrix_emp if cube_ref is None else prepare_matrix(cube_ref, scaling, mask_center_px)
    if left_eigv:                                  # temporal modes (pca_fullfr.py:1720-1724): same subspace projection
        V = svd_wrapper(ref_lib, svd_mode, ncomp, seed=seed, left_eigv=True)
        transformed = matrix_emp.T @ V
        reconstructed = V @ transformed.T
    else:
        V = svd_wrapper(ref_lib, svd_mode, ncomp, seed=seed)
        transformed = V @ matrix_emp.T
        reconstructed = transformed.T @ V
    residuals = (matrix - reconstructed).reshape(n, y, x)
    if full_output:
        return residuals, reconstructed, V
    return residuals


# --------------------------------------------------------------------------------------
# FFT 3-shear rotation (imlib='vip-fft')
# --------------------------------------------------------------------------------------


def rot_geometry(N):
    """Padded sizes for an N-pixel axis.  Ref: derotation.py:154-158 (1.5x canvas, parity
    matched), cosmetics.py:210-215 (x 4/1.5, parity matched), derotation.py:583-599.
    Returns (L, Le, off): padded period, even work length, offset of the frame in the
    canvas (pixel N//2 lands on L//2)."""
    n1 = int(N * 1.5)
    if n1 % 2 != N % 2:
        n1 += 1
    L = int(round(n1 * (4 / 1.5)))
    if L % 2 != n1 % 2:
        L -= 1
    Le = L if L % 2 == 0 else L - 1
    off = L // 2 - N // 2
    return L, Le, off


def rot_angle_split(angle):
    """Wrap to [0, 360], split into q quarter turns (np.rint: half-to-even) and residual d.
    Ref: derotation.py:577-596."""
    a = float(angle)
    while a < 0:
        a += 360
    while a > 360:
        a -= 360
    if a > 45:
        d = a % 90
        if d > 45:
            d = -(90 - d)
        q = int(np.rint(a / 90))
    else:
        d = a
        q = 0
    return q, d


def _line_shift(arr, shifts, axis):
    """Circular sinc shift (period = len along ``axis``) of every line along ``axis`` by
    ``shifts[line]`` pixels: ifft(fft(line) * exp(-2 pi i f s)).  Ref: derotation.py:625-640
    (the four full-array fftshifts cancel for even lengths)."""
    Le = arr.shape[axis]
    f = np.fft.fftfreq(Le)
    if axis == 1:
        ph = np.exp(-2j * np.pi * shifts[:, None] * f[None, :])
    else:
        ph = np.exp(-2j * np.pi * f[:, None] * shifts[None, :])
    return np.fft.ifft(np.fft.fft(arr, axis=axis) * ph, axis=axis)


def frame_rotate_fft(frame, angle, mask_val=np.nan):
    """frame_rotate(..., imlib='vip-fft', edge_blend=None) -> float64 (Ny, Nx).
    Ref: derotation.py:129-222,236-237,324-326,542-640."""
    if frame.ndim != 2:
        raise TypeError("Input array is not a frame or 2d array")
    Ny, Nx = frame.shape
    if Ny != Nx:
        raise ValueError("oracle restates the square-frame case only")
    N = Ny
    L, Le, off = rot_geometry(N)
    if np.isnan(mask_val):
        mask_ori = np.isnan(frame)
    else:
        mask_ori = frame == mask_val
    canvas = np.zeros((L + 1, L + 1))                 # odd embedding so the pivot is a pixel
    canvas[off:off + N, off:off + N] = np.where(np.isnan(frame), 0.0, frame)
    if L % 2:                                         # odd L is already odd: no embedding
        canvas = canvas[:L, :L]
    q, d = rot_angle_split(angle)
    if q:
        canvas = np.rot90(canvas, q)
    W = canvas[:Le, :Le].astype(complex)
    a = np.tan(np.deg2rad(d) / 2)
    b = -np.sin(np.deg2rad(d))
    c = L // 2
    coord = np.arange(Le) - c
    W = _line_shift(W, a * coord, axis=1)             # rows shifted along x by a*(y-c)
    W = _line_shift(W, b * coord, axis=0)             # cols shifted along y by b*(x-c)
    W = _line_shift(W, a * coord, axis=1)
    out = np.real(W)[off:off + N, off:off + N].copy()
    out[mask_ori] = mask_val
    return out


def _warp_weights(taps):
    """Separable float32 weights of OpenCV's 32 sub-pixel phases (imgwarp.cpp interpolateLinear / Cubic / Lanczos4)."""
    tab = np.zeros((32, taps), dtype=np.float32)
    for i in range(32):
        x = np.float32(i / 32.0)
        one = np.float32(1)
        if taps == 2:
            tab[i] = (one - x, x)
        elif taps == 4:
            A = np.float32(-0.75)
            c0 = ((A * (x + one) - 5 * A) * (x + one) + 8 * A) * (x + one) - 4 * A
            c1 = ((A + 2) * x - (A + 3)) * x * x + one
            c2 = ((A + 2) * (one - x) - (A + 3)) * (one - x) * (one - x) + one
            tab[i] = (c0, c1, c2, one - np.float32(c0) - np.float32(c1) - np.float32(c2))
        else:
            if x < np.finfo(np.float32).eps:
                tab[i, 3] = 1
                continue
            s45 = 0.70710678118654752440084436210485
            cs = [(1, 0), (-s45, -s45), (0, 1), (s45, -s45), (-1, 0), (s45, s45), (0, -1), (-s45, s45)]
            y0 = -(float(x) + 3) * np.pi * 0.25
            s0, c0 = np.sin(y0), np.cos(y0)
            c = np.array([(cs[k][0] * s0 + cs[k][1] * c0) / (-(float(x) + 3 - k) * np.pi * 0.25) ** 2 for k in range(8)],
                         dtype=np.float32)
            tot = np.float32(0)
            for k in range(8):
                tot = np.float32(tot + c[k])
            tab[i] = c * np.float32(one / tot)
    return tab


def _border_index(p, length, mode):
    """cv::borderInterpolate for the reference's border_mode names (derotation.py:294-305); -1 = constant border."""
    p = np.asarray(p, dtype=np.int64)
    if mode == "constant":
        return np.where((p >= 0) & (p < length), p, -1)
    if mode == "edge":
        return np.clip(p, 0, length - 1)
    if mode == "wrap":
        return np.mod(p, length)
    if length == 1:
        return np.zeros_like(p)
    period = 2 * length if mode == "symmetric" else 2 * length - 2      # fedcba|abcdefgh|hgfedcb  /  gfedcb|abcdefgh|gfedcba
    q = np.mod(p, period)
    return np.where(q < length, q, period - q - (1 if mode == "symmetric" else 0))


def warp_rotate(frame, angle, interpolation="lanczos4", cxy=None, border_mode="constant"):
    """``frame_rotate(frame, angle, imlib='opencv', interpolation=..., border_mode='constant')``.
    Ref: preproc/derotation.py:218 (NaN -> 0), :223-226 (centre = frame_center), :279-305 (cv2.getRotationMatrix2D +
    cv2.warpAffine on float32).  PARITY UNPINNED: opencv-python (pyproject.toml:56, no version pin) is not installed
    here, so this restates OpenCV 4.x's published algorithm (modules/imgproc/src/imgwarp.cpp: inverted affine map in
    double, source coordinates in 1/1024-pixel fixed point truncated to 1/32 pixel, separable float tables, border
    constant 0) and could not be compared with cv2 itself."""
    taps = {"nearneig": 1, "bilinear": 2, "bicubic": 4, "lanczos4": 8}[interpolation]
    a = np.where(np.isnan(np.asarray(frame, dtype=np.float32)), np.float32(0), np.asarray(frame, dtype=np.float32))
    ny, nx = a.shape
    cy, cx = frame_center(a) if cxy is None else (cxy[1], cxy[0])
    al, be = np.cos(np.deg2rad(angle)), np.sin(np.deg2rad(angle))
    M = np.array([al, be, (1 - al) * cx - be * cy, -be, al, be * cx + (1 - al) * cy], dtype=np.float64)
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0], M[1], M[3], M[4] = A11, M[1] * -D, M[3] * -D, A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    xs = np.arange(nx, dtype=np.float64)
    ys = np.arange(ny, dtype=np.float64)
    rd = 512 if taps == 1 else 16
    X0 = (np.rint((M[1] * ys + M[2]) * 1024).astype(np.int64) + rd)[:, None] + np.rint(M[0] * xs * 1024).astype(np.int64)[None]
    Y0 = (np.rint((M[4] * ys + M[5]) * 1024).astype(np.int64) + rd)[:, None] + np.rint(M[3] * xs * 1024).astype(np.int64)[None]
    def tap(sy, sx):
        iy, ix = _border_index(sy, ny, border_mode), _border_index(sx, nx, border_mode)
        return np.where((iy >= 0) & (ix >= 0), a[np.maximum(iy, 0), np.maximum(ix, 0)], np.float32(0))

    if taps == 1:
        return tap(Y0 >> 10, X0 >> 10)
    X, Y = X0 >> 5, Y0 >> 5
    tab = _warp_weights(taps)
    wx, wy = tab[X & 31], tab[Y & 31]                           # (ny, nx, taps)
    sx, sy = (X >> 5) - (taps // 2 - 1), (Y >> 5) - (taps // 2 - 1)
    out = np.zeros((ny, nx), dtype=np.float32)
    for r in range(taps):
        for c in range(taps):
            out += tap(sy + r, sx + c) * (wy[..., r] * wx[..., c])
    return out


def cube_derotate(cube, angle_list, mask_val=np.nan, out_dtype=None):
    """nproc=1 semantics: output dtype = input dtype.  Ref: derotation.py:383-391."""
    if cube.ndim != 3:
        raise TypeError("Input array is not a cube or 3d array.")
    out = np.zeros_like(cube) if out_dtype is None else np.zeros(cube.shape, out_dtype)
    for i in range(cube.shape[0]):
        out[i] = frame_rotate_fft(cube[i], -angle_list[i], mask_val=mask_val)
    return out


# --------------------------------------------------------------------------------------
# collapse
# --------------------------------------------------------------------------------------


def cube_collapse(cube, mode="median", n=50, w=None):
    """3-D: axis 0, 4-D: axis 1.  numpy nan-functions (bottleneck absent == numpy
    semantics).  Ref: preproc/subsampling.py:30-116."""
    if cube.ndim == 3:
        ax = 0
    elif cube.ndim == 4:
        ax = 1
    else:
        raise TypeError("The input array is not a cube or 3d array.")
    if mode == "mean":
        return np.nanmean(cube, axis=ax)
    if mode == "median":
        return np.nanmedian(cube, axis=ax)
    if mode == "sum":
        return np.nansum(cube, axis=ax)
    if mode == "max":
        return np.nanmax(cube, axis=ax)
    if mode == "absmean":
        return np.nanmean(np.abs(cube), axis=ax)
    if mode == "wmean":
        if w is None:
            raise ValueError("Weights have to be provided for weighted mean mode")
        if len(w) != cube.shape[0]:
            raise TypeError("Weights need same length as cube")
        arr = np.where(np.isnan(cube), 0, cube)
        if ax == 0:
            return np.inner(np.asarray(w), np.moveaxis(arr, 0, -1))
        return np.stack([np.inner(np.asarray(w), np.moveaxis(arr[j], 0, -1))
                         for j in range(arr.shape[0])])
    if mode == "trimmean":
        if ax != 0:
            raise NotImplementedError
        N = cube.shape[0]
        k = (N - n) // 2
        if N % 2 != n % 2:
            n += 1
        srt = np.sort(cube, axis=0)
        return np.nanmean(srt[k:k + n], axis=0).astype(cube.dtype)
    raise TypeError("mode not recognized")


# --------------------------------------------------------------------------------------
# ADI index helpers (bit-exact contracts)
# --------------------------------------------------------------------------------------


def find_indices_adi(angle_list, frame, thr, truncate=False, max_frames=200, nframes=None):
    """Library indices for ``frame``.  Ref: preproc/derotation.py:410-496 (out_closest=False; ``nframes``: the
    nframes/2 frames on either side of the excluded window, :458-474)."""
    n = angle_list.shape[0]
    index_prev = 0
    for i in range(frame):
        if abs(angle_list[frame] - angle_list[i]) < thr:
            break
        index_prev += 1
    index_foll = frame
    for k in range(frame, n):
        if abs(angle_list[k] - angle_list[frame]) > thr:
            break
        index_foll += 1
    if nframes is not None:
        window = nframes // 2
        ind1, ind4 = max(index_prev - window, 0), min(index_foll + window, n)
        return np.array(list(range(ind1, index_prev)) + list(range(index_foll, ind4)), dtype="int32")
    idx = np.array(list(range(0, index_prev)) + list(range(index_foll, n)), dtype="int32")
    if truncate:
        lim = min(n - 1, max_frames)
        if len(idx) > lim:
            allidx = np.array(list(range(0, index_prev)) + list(range(index_foll, n)))
            dpa = np.abs(angle_list[allidx] - angle_list[frame])
            idx = np.sort(allidx[np.argsort(dpa)][:lim])
    return idx


def compute_pa_thresh(ann_center, fwhm, delta_rot=1):
    """Ref: preproc/derotation.py:499-504."""
    return np.rad2deg(2 * np.arctan(delta_rot * fwhm / (2 * ann_center)))


def define_annuli(angle_list, ann, n_annuli, fwhm, radius_int, annulus_width, delta_rot,
                  strict=True):
    """Ref: preproc/derotation.py:507-539 (prints dropped)."""
    if ann == n_annuli - 1:
        inner_radius = radius_int + (ann * annulus_width - 1)
    else:
        inner_radius = radius_int + ann * annulus_width
    ann_center = inner_radius + (annulus_width / 2)
    pa_threshold = compute_pa_thresh(ann_center, fwhm, delta_rot)
    mid_range = np.abs(np.amax(angle_list) - np.amin(angle_list)) / 2
    if pa_threshold >= mid_range - mid_range * 0.1 and not strict:
        pa_threshold = float(mid_range - mid_range * 0.1)
    return pa_threshold, inner_radius, ann_center


def get_annulus_segments(shape, inner_radius, width, nsegm=1, theta_init=0):
    """mode='ind', optim_scale_fact=1.  Ref: var/shapes.py:474-581."""
    if not isinstance(nsegm, int):
        raise TypeError("`nsegm` must be an integer")
    cy, cx = frame_center(shape)
    az = np.deg2rad(int(np.ceil(360 / nsegm)))
    twopi = 2 * np.pi
    yy, xx = np.mgrid[:shape[0], :shape[1]]
    rad = np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2)
    phirot = np.arctan2(yy - cy, xx - cx) % twopi
    outer = inner_radius + width
    ring = (rad >= inner_radius) & (rad < outer)
    res = []
    for i in range(nsegm):
        ps = np.deg2rad(theta_init) + i * az
        pe = ps + az
        if ps < twopi and pe > twopi:
            m = ring & (phirot >= ps) & (phirot <= twopi) | ring & (phirot >= 0) & (phirot < pe - twopi)
        elif ps >= twopi and pe > twopi:
            m = ring & (phirot >= ps - twopi) & (phirot < pe - twopi)
        else:
            m = ring & (phirot >= ps) & (phirot < pe)
        res.append(np.where(m))
    return res


# --------------------------------------------------------------------------------------
# end-to-end entry points
# --------------------------------------------------------------------------------------


def pca_fullframe(cube, angle_list, ncomp=1, svd_mode="lapack", scaling=None,
                  mask_center_px=None, collapse="median", cube_ref=None, weights=None,
                  full_output=False, seed=0, rot_options=None, cube_sig=None, left_eigv=False):
    """3-D ADI / RDI branch of ``pca``.  Ref: psfsub/pca_fullfr.py:412-415,661-701,
    801-1007,759-793; ``left_eigv`` (ADI only, :428-437): pcs = V.T, (k x n) (:905)."""
    if left_eigv and cube_ref is not None:
        raise NotImplementedError("left_eigv is not compatible with 'mask_rdi' nor 'batch'")
    if cube.ndim != 3:
        raise TypeError("`cube` must be a 3d numpy ndarray")
    n = cube.shape[0]
    rot_options = dict(rot_options or {})
    if mask_center_px and len(rot_options) == 0:
        rot_options = {"mask_val": 0, "ker": 1, "interp_zeros": True}
    mask_val = rot_options.get("mask_val", np.nan)
    angle_list = check_pa_vector(np.asarray(angle_list, dtype=float))
    if n != angle_list.shape[0]:
        raise ValueError("`angle_list` vector has wrong length. It must equal the number of "
                         "frames in the cube")
    nref = n if cube_ref is None else cube_ref.shape[0]
    if isinstance(ncomp, (int, np.integer)) and ncomp > nref:
        ncomp = min(ncomp, nref)
    elif ncomp <= 0:
        raise ValueError("Number of PCs too low. It should be > 0.")
    res, recon, V = project_subtract(cube, ncomp, scaling, mask_center_px, svd_mode,
                                     cube_ref=cube_ref, full_output=True, seed=seed, cube_sig=cube_sig,
                                     left_eigv=left_eigv)
    y, x = cube.shape[1:]
    pcs = V.T if left_eigv else V.reshape(V.shape[0], y, x)
    recon = recon.reshape(n, y, x)
    res_der = cube_derotate(res, angle_list, mask_val=mask_val)
    frame = cube_collapse(res_der, mode=collapse, w=weights)
    if mask_center_px:
        res_der = mask_circle(res_der, mask_center_px)
        frame = mask_circle(frame, mask_center_px)
    if full_output:
        return frame, pcs, recon, res, res_der
    return frame


def pca_grid_frames(cube, angle_list, range_pcs, scaling=None, mask_center_px=None, collapse="median",
                    cube_ref=None, weights=None, svd_mode="lapack", full_output=False, seed=0):
    """``pca(cube, angles, ncomp=<tuple or list>)`` without ``source_xy``: the grid of final frames.
    Ref: psfsub/pca_fullfr.py:412-415,1010-1035 -> psfsub/utils_pca.py:131-161 (truncate_svd_get_finframe),
    :241-300 (pclist, one decomposition with pcmax), :418-428 (returns)."""
    n, y, x = cube.shape
    mask_val = 0 if mask_center_px else np.nan          # pca(): mask_center_px without rot_options -> mask_val=0
    angle_list = check_pa_vector(np.asarray(angle_list, dtype=float))
    if isinstance(range_pcs, list):
        pclist = list(range_pcs)
        pcmax = max(pclist)
    else:
        if len(range_pcs) == 2:
            pcmin, pcmax = range_pcs
            pcmax = min(pcmax, n)
            step = 1
        elif len(range_pcs) == 3:
            pcmin, pcmax, step = range_pcs
            pcmax = min(pcmax, n)
        else:
            raise TypeError("`range_pcs` must be None or a tuple")
        pclist = list(range(pcmin, pcmax + 1, step))
    matrix = prepare_matrix(cube, scaling, mask_center_px)
    ref_lib = prepare_matrix(cube_ref, scaling, mask_center_px) if cube_ref is not None else matrix
    V = svd_wrapper(ref_lib, svd_mode, pcmax, seed=seed)
    frames = []
    for pc in pclist:
        transformed = np.dot(V[:pc], matrix.T)
        reconstructed = np.dot(transformed.T, V[:pc])
        residuals = (matrix - reconstructed).reshape(n, y, x)
        der = cube_derotate(residuals, angle_list, mask_val=mask_val)
        frames.append(cube_collapse(der, mode=collapse, w=weights))
    cubeout = np.array(frames)
    if full_output:
        return cubeout, pclist
    return cubeout


# --------------------------------------------------------------------------------------
# S/N of a test aperture (metrics/snr_source.py) -- PARITY UNPINNED for the aperture sums: the reference takes them from
# photutils.aperture_photometry(method='exact') (photutils 2.3.0 in its uv.lock; absent from this image), i.e. every
# pixel weighted by the exact area of its unit square inside the circle.  Restated here by 1-D quadrature of that area
# (independent of the product's closed form); everything around it (aperture centres, Student-t statistic, grid
# scoring) follows the reference's lines.
# --------------------------------------------------------------------------------------

_GL_X, _GL_W = np.polynomial.legendre.leggauss(48)


def _pixel_circle_area(x0, y0, r):
    """Area of the pixel [x0, x0+1] x [y0, y0+1] (relative to the circle centre) inside the circle of radius r:
    integral over u of the length of [y0, y0+1] n [-h(u), h(u)], h = sqrt(r^2 - u^2), with u = r sin(t) (removes the
    square-root end-point singularity) and Gauss-Legendre on every smooth piece."""
    a, b = max(x0, -r), min(x0 + 1.0, r)
    if b <= a:
        return 0.0
    cuts = {a, b}
    for yv in (y0, y0 + 1.0):
        if abs(yv) < r:
            for u in (np.sqrt(r * r - yv * yv), -np.sqrt(r * r - yv * yv)):
                if a < u < b:
                    cuts.add(float(u))
    cuts = sorted(cuts)
    total = 0.0
    for ca, cb in zip(cuts[:-1], cuts[1:]):
        ta, tb = np.arcsin(np.clip(ca / r, -1, 1)), np.arcsin(np.clip(cb / r, -1, 1))
        t = 0.5 * (tb - ta) * _GL_X + 0.5 * (tb + ta)
        h = r * np.cos(t)
        seg = np.maximum(0.0, np.minimum(y0 + 1.0, h) - np.maximum(y0, -h))
        total += 0.5 * (tb - ta) * np.sum(_GL_W * seg * r * np.cos(t))
    return float(total)


def aperture_sum_exact(array, xc, yc, r):
    """photutils CircularAperture((xc, yc), r) summed with method='exact' (pixel centres at integer coordinates)."""
    ny, nx = array.shape
    tot = 0.0
    for j in range(max(int(np.floor(yc - r + 0.5)), 0), min(int(np.ceil(yc + r - 0.5)), ny - 1) + 1):
        for i in range(max(int(np.floor(xc - r + 0.5)), 0), min(int(np.ceil(xc + r - 0.5)), nx - 1) + 1):
            tot += _pixel_circle_area(i - 0.5 - xc, j - 0.5 - yc, r) * float(array[j, i])
    return tot


def indep_ap_centers(shape, source_xy, fwhm, exclude_negative_lobes=False):
    """metrics/snr_source.py:226-318 (no exclude_theta_range / no_gap): (yy, xx), the test aperture first."""
    sourcex, sourcey = source_xy
    centery, centerx = frame_center(shape)
    sep = np.sqrt((centery - float(sourcey)) ** 2 + (centerx - float(sourcex)) ** 2)
    if not sep > (fwhm / 2):
        raise RuntimeError("`source_xy` is too close to the frame center")
    angle = np.arcsin(fwhm / 2.0 / sep) * 2
    nap = int(np.floor(2 * np.pi / angle))
    ca, sa = np.cos(angle), np.sin(angle)
    xs, ys = [sourcex - centerx], [sourcey - centery]
    xa, ya = np.zeros(nap), np.zeros(nap)
    xa[0], ya[0] = xs[0], ys[0]
    for i in range(nap - 1):                          # clockwise: sign = -1
        xa[i + 1] = ca * xa[i] + sa * ya[i]
        ya[i + 1] = ca * ya[i] - sa * xa[i]
        if exclude_negative_lobes and (i == 0 or i == nap - 2):
            continue
        xs.append(xa[i + 1])
        ys.append(ya[i + 1])
    return np.array(ys) + centery, np.array(xs) + centerx


def snr(array, source_xy, fwhm, full_output=False, exclude_negative_lobes=False):
    """metrics/snr_source.py:321-456 (one array)."""
    yy, xx = indep_ap_centers(array.shape, source_xy, fwhm, exclude_negative_lobes)
    fluxes = np.array([aperture_sum_exact(array, x_, y_, fwhm / 2.0) for y_, x_ in zip(yy, xx)])
    f_source = fluxes[0]
    rest = fluxes[1:]
    n2 = rest.shape[0]
    val = (f_source - rest.mean()) / (rest.std(ddof=1) * np.sqrt(1 + (1 / n2)))
    if full_output:
        return source_xy[1], source_xy[0], f_source, rest, val
    return val


def pca_grid_snr(cube, angle_list, range_pcs, source_xy, fwhm, fmerit="mean", **kw):
    """``pca(cube, angles, ncomp=<tuple/list>, source_xy=..., fwhm=...)``: grid of frames scored by S/N at ``source_xy``.
    Ref: psfsub/utils_pca.py:239-277 (get_snr), :364-402 (loop, argmax, table).  Returns (cubeout, finalfr, pclist,
    snrlist, fluxlist, opt_npc)."""
    cubeout, pclist = pca_grid_frames(cube, angle_list, range_pcs, full_output=True, **kw)
    x, y = source_xy
    snrlist, fluxlist = [], []
    for fr in cubeout:
        fr = np.asarray(fr, dtype=np.float64)
        if fmerit == "px":
            r = snr(fr, (x, y), fwhm, full_output=True)
            sv, fl = r[-1], r[2]
        else:
            yy, xx = disk_indices((y, x), fwhm / 2.0, None)
            res = [snr(fr, (x_, y_), fwhm, full_output=True) for y_, x_ in zip(yy, xx)]
            sn = np.array([r[-1] for r in res])
            fx = np.array([r[2] for r in res])
            sv, fl = (np.max(sn), fx[int(np.argmax(sn))]) if fmerit == "max" else (np.mean(sn), np.mean(fx))
        snrlist.append(0 if np.isnan(sv) else sv)
        fluxlist.append(fl)
    argmax = int(np.argmax(snrlist))
    return cubeout, cubeout[argmax], pclist, snrlist, fluxlist, pclist[argmax]


def pca_annulus(cube, angs, ncomp, annulus_width, r_guess, cube_ref=None, svd_mode="lapack", scaling=None,
                collapse="median", weights=None, mask_val=np.nan):
    """3-D branch of psfsub/utils_pca.py:678-709: PCA on the one-segment annulus [r_guess -+ width/2], residuals scattered
    into a zero cube, derotated, collapsed."""
    inrad = int(r_guess - annulus_width / 2.0)
    outrad = int(r_guess + annulus_width / 2.0)
    yy, xx = get_annulus_segments(cube.shape[1:], inrad, int(np.round(outrad - inrad)), 1)[0]
    data = matrix_scaling(cube[:, yy, xx], scaling)
    data_svd = data if cube_ref is None else matrix_scaling(cube_ref[:, yy, xx], scaling)
    V = svd_wrapper(data_svd, svd_mode, ncomp)
    residuals = data - np.dot(np.dot(data, V.T), V)
    cube_zeros = np.zeros_like(cube)
    cube_zeros[:, yy, xx] = residuals
    out = cube_zeros if angs is None else cube_derotate(cube_zeros, check_pa_vector(np.asarray(angs, dtype=float)),
                                                        mask_val=mask_val)
    if collapse is not None:
        return cube_collapse(out, mode=collapse, w=weights)
    return out


def pca_pa_rejection(cube, angle_list, ncomp, source_xy, fwhm, delta_rot, scaling=None, mask_center_px=None,
                     min_frames_pca=10, max_frames_pca=None, collapse="median", svd_mode="lapack", weights=None,
                     full_output=False, seed=0, cube_sig=None, cube_ref=None):
    """``pca(cube, angles, ncomp=<int>, source_xy=(x, y), fwhm=..., delta_rot=...)``: every frame is modelled with
    the PCs of the frames that rotated by more than the PA threshold at ``source_xy`` (with ``cube_ref``: plus every
    reference frame, :1693-1694).
    Ref: psfsub/pca_fullfr.py:911-965 (threshold, per-frame loop), :1677-1713 (_project_subtract with indices/frame),
    :966-991 (derotate, collapse, mask), :779-785 (returns)."""
    n, y, x = cube.shape
    mask_val = 0 if mask_center_px else np.nan
    angle_list = check_pa_vector(np.asarray(angle_list, dtype=float))
    yc, xc = frame_center(cube[0])
    x1, y1 = source_xy
    ann_center = np.sqrt((yc - y1) ** 2 + (xc - x1) ** 2)
    pa_thr = compute_pa_thresh(ann_center, fwhm, delta_rot)
    truncate = max_frames_pca is not None
    matrix = prepare_matrix(cube, scaling, mask_center_px)
    matrix_emp = matrix if cube_sig is None else matrix - np.reshape(cube_sig, (n, -1))     # :1652-1662
    residuals = np.zeros_like(matrix)
    recon = np.zeros_like(matrix)
    matrix_ref = None if cube_ref is None else prepare_matrix(cube_ref, scaling, mask_center_px)
    for fr in range(n):
        ind = find_indices_adi(angle_list, fr, pa_thr, truncate=truncate, max_frames=max_frames_pca)
        ref_lib = matrix_emp[ind]
        if matrix_ref is not None:
            ref_lib = np.concatenate((ref_lib, matrix_ref))
        if ref_lib.shape[0] < min_frames_pca:
            raise RuntimeError("{} frames comply to delta_rot condition < less than min_frames_pca ({})".format(
                ref_lib.shape[0], min_frames_pca))
        if ref_lib.shape[0] < ncomp:
            raise RuntimeError("{} frames comply to delta_rot condition < less than ncomp ({})".format(
                ref_lib.shape[0], ncomp))
        V = svd_wrapper(ref_lib, svd_mode, ncomp, seed=seed)
        transformed = np.dot(matrix_emp[fr], V.T)
        recon[fr] = np.dot(transformed.T, V)
        residuals[fr] = matrix[fr] - recon[fr]
    res_cube = residuals.reshape(n, y, x)
    der = cube_derotate(res_cube, angle_list, mask_val=mask_val)
    frame = cube_collapse(der, mode=collapse, w=weights)
    if mask_center_px:
        der = mask_circle(der, mask_center_px)
        frame = mask_circle(frame, mask_center_px)
    if full_output:
        return frame, recon.reshape(n, y, x), res_cube, der
    return frame


def median_sub_fullfr(cube, angle_list, radius_int=0, collapse="median", cube_ref=None, collapse_ref="median",
                      full_output=False):
    """``median_sub(cube, angles, mode='fullfr')`` for a 3-D cube.  Ref: psfsub/medsub.py:226-229 (mask default),
    :246-253 (reference frame), :279-281 (median model), :288-319 (full-frame), :376-387 (derotate, mask, collapse),
    :516-519 (returns)."""
    arr = cube.copy()
    mask_val = 0 if radius_int else np.nan
    angle_list = check_pa_vector(np.asarray(angle_list, dtype=float))
    if cube_ref is not None:
        ref_frame = np.median(cube_ref, axis=0) if "median" in collapse_ref else np.mean(cube_ref, axis=0)
        arr -= ref_frame
    else:
        arr -= np.median(arr, axis=0)
    cube_out = arr
    cube_der = cube_derotate(cube_out, angle_list, mask_val=mask_val)
    if radius_int:
        cube_out = mask_circle(cube_out, radius_int)
        cube_der = mask_circle(cube_der, radius_int)
    frame = cube_collapse(cube_der, mode=collapse)
    if full_output:
        return cube_out, cube_der, frame
    return frame


def median_sub_annular(cube, angle_list, fwhm=4, radius_int=0, asize=4, delta_rot=1, nframes=4, collapse="median",
                       cube_ref=None, collapse_ref="median", full_output=False):
    """``median_sub(cube, angles, mode='annular')`` for a 3-D cube.  Ref: psfsub/medsub.py:279-281 (global median
    model, ADI only), :316-371 (annuli loop, nframes must be even), :602-641 (_median_subt_ann_adi: per frame the median
    of the ``nframes`` frames closest in time beyond the PA threshold; _define_annuli with strict=False),
    :644-676 (_median_subt_ann_rdi: the collapsed reference frame, last-annulus rule NOT applied), :376-387."""
    arr = cube.copy()
    n, y, x = arr.shape
    mask_val = 0 if radius_int else np.nan
    angle_list = check_pa_vector(np.asarray(angle_list, dtype=float))
    if cube_ref is not None:
        ref_frame = np.median(cube_ref, axis=0) if "median" in collapse_ref else np.mean(cube_ref, axis=0)
    else:
        arr -= np.median(arr, axis=0)
        if nframes is not None and nframes % 2 != 0:
            raise TypeError("`nframes` argument must be even value")
    cube_out = np.zeros_like(arr)
    n_annuli = int((y / 2 - radius_int) / asize)
    for ann in range(n_annuli):
        if cube_ref is not None:
            inner_radius = radius_int + ann * asize
            yy, xx = get_annulus_segments((y, x), inner_radius, asize)[0]
            cube_out[:, yy, xx] = arr[:, yy, xx] - ref_frame[yy, xx]
            continue
        pa_thr, inner_radius, _ = define_annuli(angle_list, ann, n_annuli, fwhm, radius_int, asize, delta_rot,
                                                strict=False)
        yy, xx = get_annulus_segments((y, x), inner_radius, asize)[0]
        matrix = arr[:, yy, xx]
        res = np.zeros_like(matrix)
        for fr in range(n):
            disc = matrix[find_indices_adi(angle_list, fr, pa_thr, nframes=nframes)] if pa_thr != 0 else matrix
            with np.errstate(all="ignore"):
                res[fr] = matrix[fr] - np.nanmedian(disc, axis=0)
        cube_out[:, yy, xx] = res
    cube_der = cube_derotate(cube_out, angle_list, mask_val=mask_val)
    if radius_int:
        cube_out = mask_circle(cube_out, radius_int)
        cube_der = mask_circle(cube_der, radius_int)
    frame = cube_collapse(cube_der, mode=collapse)
    if full_output:
        return cube_out, cube_der, frame
    return frame


def stim_map(cube_der):
    """Ref: metrics/stim.py:24-44 (+ var/shapes.py:389-398 get_circle)."""
    t, n, _ = cube_der.shape
    mu = np.mean(cube_der, axis=0)
    sigma = np.sqrt(np.var(cube_der, axis=0))
    det = np.divide(mu, sigma, out=np.zeros_like(mu), where=sigma != 0)
    cy, cx = frame_center(det)
    yy, xx = np.ogrid[:det.shape[0], :det.shape[1]]
    return det * ((yy - cy) ** 2 + (xx - cx) ** 2 < int(np.round(n / 2.)) ** 2)


def inverse_stim_map(cube, angle_list):
    """Ref: metrics/stim.py:47-72."""
    return stim_map(cube_derotate(cube, -np.asarray(angle_list)))


def normalized_stim_map(cube, angle_list, mask=None):
    """Ref: metrics/stim.py:75-118."""
    inv = inverse_stim_map(cube, angle_list)
    if mask is not None:
        inv = mask_circle(inv, mask) if np.isscalar(mask) else inv * mask
    max_inv = np.nanmax(inv)
    if max_inv <= 0:
        raise ValueError("The normalization value is found to be {}".format(max_inv))
    return stim_map(cube_derotate(cube, angle_list)) / max_inv


# --------------------------------------------------------------------------------------
# ADI+mSDI: FFT rescaling of the spectral channels and the single / double pass PCA
# --------------------------------------------------------------------------------------
def scale_fft(array, scale, ori_dim=False):
    """FFT zoom of an even square frame.  Ref: preproc/rescaling.py:1114-1217."""
    if scale == 1:
        return array
    dim = array.shape[0]
    kd_array = np.arange(dim / 2 + 1, dtype=int)
    yy = dim / 2 * (scale - 1) + kd_array.astype(float) * scale
    kf_array = np.round(yy).astype(int)
    imin = np.nanargmin(np.abs(yy - kf_array))
    kd_io, kf_io = kd_array[imin], kf_array[imin]
    dim_p = int(dim + 2 * kd_io)
    tmp = np.zeros((dim_p, dim_p), dtype=array.dtype.kind)
    tmp[kd_io:kd_io + dim, kd_io:kd_io + dim] = array
    array_f = np.fft.fftshift(np.fft.fft2(tmp))
    dim_pp = int(dim + 2 * kf_io)
    if dim_pp > dim_p:
        tmp = np.zeros((dim_pp, dim_pp), dtype=complex)
        lo, hi = (dim_pp - dim_p) // 2, (dim_pp + dim_p) // 2
        tmp[lo:hi, lo:hi] = array_f
    else:
        tmp = array_f[kd_io - kf_io:kd_io - kf_io + dim_pp, kd_io - kf_io:kd_io - kf_io + dim_pp]
    array_resc = np.fft.ifft2(np.fft.fftshift(tmp)).real
    dim_resc = int(round(scale * dim))
    if dim_resc > dim and dim_resc % 2 != dim % 2:
        dim_resc += 1
    elif dim_resc < dim and dim_resc % 2 != dim % 2:
        dim_resc -= 1
    if not ori_dim and dim_pp > dim_resc:
        lo, hi = (dim_pp - dim_resc) // 2, (dim_pp + dim_resc) // 2
        array_resc = array_resc[lo:hi, lo:hi]
    elif not ori_dim and dim_pp <= dim_resc:
        out = np.zeros((dim_resc, dim_resc))
        lo, hi = (dim_resc - dim_pp) // 2, (dim_resc + dim_pp) // 2
        out[lo:hi, lo:hi] = array_resc
        array_resc = out
    elif dim_pp > dim:
        array_resc = array_resc[kf_io:kf_io + dim, kf_io:kf_io + dim]
    elif dim_pp <= dim:
        scaled = array * 0
        scaled[-kf_io:-kf_io + dim_pp, -kf_io:-kf_io + dim_pp] = array_resc
        array_resc = scaled
    return array_resc


def frame_rescaling_fft(array, scale):
    """frame_rescaling(imlib='vip-fft') about the frame centre, NaN-free input.
    Ref: preproc/rescaling.py:560-575 (centre check), :636-672 (vip-fft branch, odd frames embedded at [1:, 1:])."""
    if scale is None:
        scale = 1.0
    if array.shape[0] != array.shape[1]:
        raise ValueError("FFT scaling only supports square input arrays")
    if array.shape[0] % 2:
        even = np.zeros([array.shape[0] + 1, array.shape[1] + 1])
        even[1:, 1:] = array
        return scale_fft(even, scale, ori_dim=True)[1:, 1:]
    return scale_fft(array, scale, ori_dim=True)


def get_square(array, size, y, x):
    """Ref: var/shapes.py:290-352 (force=False)."""
    size_init = array.shape[0]
    if size >= array.shape[0] and size >= array.shape[1]:
        raise ValueError("`Size` is equal to or bigger than the initial frame size")
    if size_init % 2 == 0:
        if size % 2 != 0:
            size += 1
    elif size % 2 == 0:
        size += 1
    wing = (size - 1) / 2
    y0, y1 = int(y - wing), int(y + wing + 1)
    x0, x1 = int(x - wing), int(x + wing + 1)
    if y0 < 0 or x0 < 0 or y1 > array.shape[0] or x1 > array.shape[1]:
        raise RuntimeError("square cannot be obtained with size={}, y={}, x={}".format(size, y, x))
    return array[y0:y1, x0:x1].copy()


def cube_crop_frames(array, size):
    """Centre crop of the frames of a 3-D cube.  Ref: preproc/cosmetics.py:66-109 (xy=None, force=False)."""
    fr = array[0]
    if fr.shape[0] == size and fr.shape[1] == size:
        return array
    cy, cx = frame_center(fr)
    wing = ((size + (1 if (fr.shape[0] % 2 == 0) != (size % 2 == 0) else 0)) - 1) / 2
    y0, x0 = int(cy - wing), int(cx - wing)
    if fr.shape[0] % 2 == 0:
        if size % 2 != 0:
            size += 1
    elif size % 2 == 0:
        size += 1
    return array[:, y0:y0 + size, x0:x0 + size]


def cube_rescaling_wavelengths(cube, scal_list, full_output=True, inverse=False, y_in=None, x_in=None,
                               collapse="median", pad_mode="reflect"):
    """Ref: preproc/rescaling.py:427-475 (imlib='vip-fft')."""
    n, y, x = cube.shape
    scal_list = np.asarray(scal_list, dtype=float)
    max_sc = np.amax(scal_list)
    if not inverse and max_sc > 1:
        new_y, new_x = int(np.ceil(max_sc * y)), int(np.ceil(max_sc * x))
        if (new_y - y) % 2 != 0:
            new_y += 1
        if (new_x - x) % 2 != 0:
            new_x += 1
        py, px = (new_y - y) // 2, (new_x - x) // 2
        big_cube = np.pad(cube, ((0, 0), (py, py), (px, px)), pad_mode)
    else:
        big_cube = cube.copy()
    n, y, x = big_cube.shape
    cy, cx = frame_center(big_cube[0])
    if inverse:
        scal_list = 1.0 / scal_list
        cy, cx = frame_center(cube[0])
    out = np.array([frame_rescaling_fft(big_cube[i], scal_list[i]) for i in range(n)])
    frame = cube_collapse(out, collapse)
    if inverse and max_sc > 1:
        if y_in is None or x_in is None:
            raise ValueError("Provide y_in and x_in when inverse=True")
        siz = max(y_in, x_in)
        if frame.shape[0] > siz:
            frame = get_square(frame, siz, cy, cx)
        if full_output and out.shape[-1] > siz:
            old = out.copy()
            out = np.zeros([old.shape[0], siz, siz])
            for zz in range(old.shape[0]):
                out[zz] = get_square(old[zz], siz, cy, cx)
    if full_output:
        return out, frame, y, x, cy, cx
    return frame


def pca_adimsdi_double(cube, angle_list, scale_list, ncomp, scaling=None, mask_center_px=None, svd_mode="lapack",
                       collapse="median", collapse_ifs="mean", ifs_collapse_range="all", weights=None,
                       full_output=False, cube_ref=None, ref_strategy="RSDI", source_xy=None, delta_rot=None, fwhm=4,
                       min_frames_pca=10, max_frames_pca=None, cube_sig=None):
    """``pca(cube4d, angles, scale_list=..., adimsdi='double', ncomp=(k_ifs, k_adi))``.
    Ref: psfsub/pca_fullfr.py:412-415 (mask default), :478-508 (routing), :1263-1549 (_adimsdi_doublepca and
    _adimsdi_doublepca_ifs), :726-731 (returns).  ``cube_ref`` (:1279-1283): its frames pass the spectral stage with
    the science frames and are the library of the second stage ('RSDI', :1388-1400) or join every frame's library
    under a rotation threshold at ``source_xy`` (:1403-1459)."""
    n_sci = cube.shape[1]
    nr = 0
    if cube_ref is not None:
        nr = cube_ref.shape[1]
        cube = np.concatenate((cube, cube_ref), axis=1)
    z, n, y_in, x_in = cube.shape
    if not isinstance(ncomp, tuple):
        raise TypeError("`ncomp` must be a tuple when a double pass PCA is performed")
    ncomp_ifs, ncomp_adi = ncomp
    mask_val = 0 if mask_center_px else np.nan
    angle_list = check_pa_vector(np.asarray(angle_list, dtype=float))
    scale_list = np.asarray(scale_list, dtype=float)
    if type(scaling) is not tuple:
        scaling = (scaling, scaling)
    if ncomp_ifs is not None and ncomp_ifs > z:
        ncomp_ifs = min(ncomp_ifs, z)
    i0, i1 = (0, z) if ifs_collapse_range == "all" else ifs_collapse_range
    res = []
    for fr in range(n):
        ms = cube[:, fr]
        if ncomp_ifs is None:
            frame_i = cube_collapse(ms[i0:i1])
        else:
            cube_resc = cube_rescaling_wavelengths(ms, scale_list)[0]
            residuals = project_subtract(cube_resc, ncomp_ifs, scaling[0], mask_center_px, svd_mode)
            frame_i = cube_rescaling_wavelengths(residuals[i0:i1], scale_list[i0:i1], full_output=False, inverse=True,
                                                 y_in=y_in, x_in=x_in, collapse=collapse_ifs)
            if mask_center_px:
                frame_i = mask_circle(frame_i, mask_center_px)
        res.append(frame_i)
    res_cube_channels = np.array(res)
    if ncomp_adi is None:
        der = cube_derotate(res_cube_channels[:n_sci], angle_list, mask_val=mask_val)
    else:
        if ncomp_adi > n:
            ncomp_adi = n
        sci, refc = res_cube_channels[:n_sci], (res_cube_channels[n_sci:] if nr else None)
        if source_xy is not None:
            res_ifs_adi = pca_pa_rejection(sci, angle_list, ncomp_adi, source_xy, fwhm, delta_rot, scaling[1],
                                           mask_center_px, min_frames_pca, max_frames_pca, svd_mode=svd_mode,
                                           full_output=True, cube_ref=refc, cube_sig=cube_sig)[2]
        elif nr and "A" not in ref_strategy:
            res_ifs_adi = project_subtract(sci, ncomp_adi, scaling[1], mask_center_px, svd_mode, cube_ref=refc,
                                           cube_sig=cube_sig)
        elif nr:
            raise IndexError("the reference de-rotates n + nr residual frames with n angles here")
        else:
            res_ifs_adi = project_subtract(res_cube_channels, ncomp_adi, scaling[1], mask_center_px, svd_mode,
                                           cube_sig=cube_sig)
        der = cube_derotate(res_ifs_adi, angle_list, mask_val=mask_val)
    frame = cube_collapse(der, mode=collapse, w=weights)
    if full_output:
        return frame, res_cube_channels, der
    return frame


def _msdi_big_cube(cube, scale_list, crop_ifs):
    """All channels of every multispectral frame rescaled, frame-major (pca_fullfr.py:1084-1096 / :1099-1115)."""
    z, n, y_in, x_in = cube.shape
    big = []
    for i in range(n):
        cr = cube_rescaling_wavelengths(cube[:, i], scale_list)[0]
        if crop_ifs:
            cr = cube_crop_frames(cr, y_in)
        big.append(cr)
    big = np.array(big)
    return big.reshape(z * n, big.shape[2], big.shape[3])


def pca_adimsdi_single(cube, angle_list, scale_list, ncomp, scaling=None, mask_center_px=None, svd_mode="lapack",
                       collapse="median", collapse_ifs="mean", ifs_collapse_range="all", crop_ifs=True, weights=None,
                       full_output=False, cube_ref=None):
    """``pca(cube4d, angles, scale_list=..., adimsdi='single', ncomp=<int | tuple | list>)``.
    Ref: psfsub/pca_fullfr.py:1038-1216 (_adimsdi_singlepca: int ncomp, optional 4-D ``cube_ref`` rescaled the same way
    and used as the PCA library, :1099-1115), :1202-1243 + psfsub/utils_pca.py:201-227 (tuple / list ncomp: grid of frames,
    every residual cube de-scaled and collapsed with ``collapse`` -- not ``collapse_ifs`` --, no central mask after the
    derotation; ``cube_ref`` is not passed on), :736-742 (returns)."""
    z, n, y_in, x_in = cube.shape
    mask_val = 0 if mask_center_px else np.nan
    angle_list = check_pa_vector(np.asarray(angle_list, dtype=float))
    scale_list = np.asarray(scale_list, dtype=float)
    big = _msdi_big_cube(cube, scale_list, crop_ifs)
    i0, i1 = (0, z) if ifs_collapse_range == "all" else ifs_collapse_range
    if isinstance(ncomp, (tuple, list)):
        nf = big.shape[0]
        if isinstance(ncomp, list):
            pclist = list(ncomp)
            pcmax = max(pclist)
        else:
            pcmin, pcmax = ncomp[0], min(ncomp[1], nf)
            pclist = list(range(pcmin, pcmax + 1, ncomp[2] if len(ncomp) == 3 else 1))
        matrix = prepare_matrix(big, scaling, mask_center_px)
        V = svd_wrapper(matrix, svd_mode, pcmax)
        frames = []
        for pc in pclist:
            res = (matrix - np.dot(np.dot(V[:pc], matrix.T).T, V[:pc])).reshape(big.shape)
            resh = np.zeros((n, y_in, y_in))
            for i in range(n):
                resh[i] = cube_rescaling_wavelengths(res[i * z + i0:i * z + i1], scale_list[i0:i1], full_output=False,
                                                     inverse=True, y_in=y_in, x_in=x_in, collapse=collapse)
            der = cube_derotate(resh, angle_list, mask_val=mask_val)
            frames.append(cube_collapse(der, mode=collapse, w=weights))
        cubeout = np.array(frames)
        return (cubeout, pclist) if full_output else cubeout
    big_ref = _msdi_big_cube(cube_ref, scale_list, crop_ifs) if cube_ref is not None else None
    res_cube = project_subtract(big, ncomp, scaling, mask_center_px, svd_mode, cube_ref=big_ref)
    resadi = np.zeros((n, y_in, x_in))
    i0, i1 = (0, z) if ifs_collapse_range == "all" else ifs_collapse_range
    desc = np.zeros_like(cube[i0:i1])
    for i in range(n):
        r = cube_rescaling_wavelengths(res_cube[i * z + i0:i * z + i1], scale_list[i0:i1], full_output=True,
                                       inverse=True, y_in=y_in, x_in=x_in, collapse=collapse_ifs)
        desc[:, i] = r[0]
        resadi[i] = r[1]
    der = cube_derotate(resadi, angle_list, mask_val=mask_val)
    if mask_center_px:
        der = mask_circle(der, mask_center_px)
    frame = cube_collapse(der, mode=collapse, w=weights)
    if full_output:
        return frame, res_cube, desc, resadi
    return frame


def pca_4d(cube, angle_list, ncomp=1, collapse_ifs="mean", full_output=False, **kw):
    """4-D cube, scale_list=None: per-channel full-frame PCA then spectral collapse.
    Ref: psfsub/pca_fullfr.py:544-658,770-774."""
    nch, nz, ny, nx = cube.shape
    ncomps = ncomp if isinstance(ncomp, list) else [ncomp] * nch
    ifs = np.zeros([nch, ny, nx])
    pcs, recon, res, resd = [], [], [], []
    for ch in range(nch):
        out = pca_fullframe(cube[ch], angle_list, ncomp=ncomps[ch], full_output=True, **kw)
        ifs[ch] = out[0]
        pcs.append(out[1]); recon.append(out[2]); res.append(out[3]); resd.append(out[4])
    frame = cube_collapse(ifs, mode=collapse_ifs)
    if full_output:
        return frame, np.array(pcs), np.array(recon), np.array(res), np.array(resd), ifs
    return frame


def pca_annular(cube, angle_list, radius_int=0, fwhm=4, asize=4, n_segments=1,
                delta_rot=(0.1, 1), ncomp=1, svd_mode="lapack", min_frames_lib=2,
                max_frames_lib=200, scaling=None, collapse="median", theta_init=0,
                weights=None, full_output=False, rot_options=None, cube_ref=None, cube_sig=None):
    """3-D ADI annular PCA (int / per-annulus tuple ncomp / LIST of ncomp: one decomposition with max(ncomp) per
    frame and segment, residuals for every V[:k] -> 4-D float64 cube_out / cube_der and a list of frames).
    Ref: psfsub/pca_local.py:228-278,594-827 (list: :665-668,744-749,799-807) and do_pca_patch :830-909
    (list: :892-902); get_eigenvectors svd.py:694-700.
    ``cube_ref`` (RDI, :716-720,879-885): the reference frames of the segment (scaled on their own) are stacked on top
    of every frame's PA-selected library.  ``cube_sig`` (:721-724,862-866,887-891): libraries and the projected frame
    are taken from ``matrix - cube_sig`` (cube_sig unscaled), the model is subtracted from ``matrix``."""
    if cube.ndim != 3:
        raise TypeError("Input array is not a cube or 3d array")
    if cube.shape[0] != np.asarray(angle_list).shape[0]:
        raise TypeError("Input vector or parallactic angles has wrong length")
    n, y, x = cube.shape
    rot_options = dict(rot_options or {})
    if radius_int and len(rot_options) == 0:
        rot_options = {"mask_val": 0, "ker": 1, "interp_zeros": True}
    mask_val = rot_options.get("mask_val", np.nan)
    angle_list = check_pa_vector(np.asarray(angle_list, dtype=float))
    n_annuli = int((y / 2 - radius_int) / asize)
    if isinstance(delta_rot, tuple):
        delta_rot = np.linspace(delta_rot[0], delta_rot[1], num=n_annuli)
    elif np.isscalar(delta_rot):
        delta_rot = [delta_rot] * n_annuli
    if isinstance(n_segments, int):
        n_segments = [n_segments] * n_annuli
    elif n_segments == "auto":
        n_segments = [2, 3]
        ld = 2 * np.tan(360 / 4 / 2) * asize          # radians quirk kept (pca_local.py:648)
        for i in range(2, n_annuli):
            ang = np.rad2deg(2 * np.arctan(ld / (2 * i * asize)))
            n_segments.append(int(np.ceil(360 / ang)))
    cube_out = np.zeros_like(cube)
    is_list = isinstance(ncomp, list)
    if is_list:
        cube_out = np.zeros([len(ncomp), n, y, x])
    for ann in range(n_annuli):
        if isinstance(ncomp, (tuple, np.ndarray)):
            if len(ncomp) != n_annuli:
                raise TypeError("If `ncomp` is a tuple, its length must match the number of annuli")
            k_ann = ncomp[ann]
        elif is_list:
            k_ann = max(ncomp)
        else:
            k_ann = ncomp
        pa_thr, inner_radius, _ = define_annuli(angle_list, ann, n_annuli, fwhm, radius_int,
                                                asize, delta_rot[ann], strict=True)
        segs = get_annulus_segments((y, x), inner_radius, asize, n_segments[ann], theta_init)
        for yy, xx in segs:
            m = matrix_scaling(cube[:, yy, xx], scaling)
            mref = matrix_scaling(cube_ref[:, yy, xx], scaling) if cube_ref is not None else None
            memp = m - cube_sig[:, yy, xx] if cube_sig is not None else m
            for fr in range(n):
                if pa_thr != 0:
                    idx = find_indices_adi(angle_list, fr, pa_thr, truncate=True,
                                           max_frames=max_frames_lib)
                    lib = memp[idx]
                    if lib.shape[0] < min_frames_lib and mref is None:
                        raise RuntimeError("Too few frames left in the PCA library.")
                else:
                    lib = memp
                if mref is not None:
                    lib = np.vstack((mref, lib))
                k = min(k_ann, min(lib.shape))
                V = svd_wrapper(lib, svd_mode, k)
                cur, cur_emp = m[fr], memp[fr]
                if is_list:
                    for nn, kk in enumerate(ncomp):
                        cube_out[nn, fr][yy, xx] = cur - (cur_emp @ V[:kk].T) @ V[:kk]
                else:
                    cube_out[fr][yy, xx] = cur - (cur_emp @ V.T) @ V
    if is_list:
        cube_der = np.zeros_like(cube_out)
        frame = []
        for nn in range(len(ncomp)):
            cube_der[nn] = cube_derotate(cube_out[nn], angle_list, mask_val=mask_val)
            frame.append(cube_collapse(cube_der[nn], mode=collapse, w=weights))
        if full_output:
            return cube_out, cube_der, frame
        return frame
    cube_der = cube_derotate(cube_out, angle_list, mask_val=mask_val)
    frame = cube_collapse(cube_der, mode=collapse, w=weights)
    if full_output:
        return cube_out, cube_der, frame
    return frame


# --------------------------------------------------------------------------------------
# synthetic ADI cubes (SURVEY.md section 8(d) generator; shared with tests/bench)
# --------------------------------------------------------------------------------------


from vip_amd.synth import synth_adi  # noqa: E402,F401  (one generator for the goldens, the tests and bench.py)
