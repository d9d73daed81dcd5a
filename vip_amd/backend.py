"""Device backend: torch tensors as device-array containers + calls into libvipmi.so.

PyTorch is plumbing only (allocation, H2D/D2H, streams); all arithmetic happens in the HIP kernels of
``vip_amd/csrc``.  One ``Context`` (vipmi_ctx) per (device, stream) is cached.
"""
import ctypes
import os
import threading

import numpy as np

from . import _lib

import collections

_ctx_cache = collections.OrderedDict()      # (device, stream handle) -> Context, least recently used first
_ctx_lock = threading.Lock()
# Every context owns hipMalloc'ed workspaces (GBs at C2 scale), so the cache is bounded: beyond MAX_CONTEXTS the least
# recently used IDLE context is dropped from the cache and trimmed (vipmi_trim: stream synchronised, workspaces freed).
# Its handle stays valid -- code that still holds the object (RcclComm.ctx, a local `ctx`) keeps working, the next
# call re-allocates; the handle itself is destroyed when the last reference goes.  A context inside a call on another
# thread is never touched (the cache grows past the bound instead).  16 >= the side streams of any one call + 1.
MAX_CONTEXTS = int(os.environ.get("VIPMI_MAX_CONTEXTS", "16"))

SCALE_MODES = {None: 0, "temp-mean": 1, "temp-standard": 2, "spat-mean": 3, "spat-standard": 4}
COLLAPSE_MODES = {"median": 0, "mean": 1, "sum": 2, "max": 3, "absmean": 4, "wmean": 5, "trimmean": 6, "stim": 7}
ROT_METHODS = {"auto": 0, "direct": 1, "fft": 2}


def _torch():
    import torch
    return torch


def require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise _lib.VipmiError("vip_amd needs an AMD MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                              "and there is no CPU fallback")
    return torch


class Context:
    def __init__(self, device=None):
        torch = require_gpu()
        self.lib = _lib.load()
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.handle = ctypes.c_void_p()
        self._in_call = threading.Lock()        # held for the duration of every library call on this context
        self._mode = (False, 0)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        st = self.lib.vipmi_create(self.device, ctypes.c_void_p(stream), ctypes.byref(self.handle))
        _lib.raise_for_status(st, "vipmi_create")
        # VIPMI_OPTS="key=value,key=value": integer tuning options applied to every new context (A/B measurements)
        for kv in filter(None, os.environ.get("VIPMI_OPTS", "").split(",")):
            k, v = kv.split("=")
            self.set_option(k.strip(), int(v))

    def bind_stream(self):
        torch = _torch()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.lib.vipmi_set_stream(self.handle, ctypes.c_void_p(stream))

    def set_option(self, key, value):
        _lib.raise_for_status(self.lib.vipmi_set_option(self.handle, key.encode(), int(value)))

    def get_option(self, key):
        return int(self.lib.vipmi_get_option(self.handle, key.encode()))

    def stage_ms(self, stage):
        return float(self.lib.vipmi_stage_ms(self.handle, stage.encode()))

    def stage_count(self, stage):
        return int(self.lib.vipmi_stage_count(self.handle, stage.encode()))

    def reset_timers(self):
        self.lib.vipmi_reset_timers(self.handle)

    def _apply_mode(self):
        """Bring the context's asynchronous-mode options in line with the calling thread's mode (set_async)."""
        mode = _mode()
        if mode != self._mode:
            on, reserve = mode
            self.set_option("eigh_check", 0 if on else 1)
            self.set_option("reserve_cus", reserve)
            self.lib.vipmi_set_gate(self.handle, _gate() if on else None)
            self._mode = mode
        if mode[0]:
            touched = getattr(_tls, "touched", None)
            if touched is None:
                touched = _tls.touched = {}
            touched[id(self)] = self

    def call(self, name, *args):
        with self._in_call:
            self._apply_mode()
            self.bind_stream()
            st = getattr(self.lib, name)(self.handle, *args)
        _lib.raise_for_status(st, name)

    def trim(self):
        """Free the workspaces if no call is running on this context (another thread); returns whether it did."""
        if not self.handle or not self._in_call.acquire(blocking=False):
            return False
        try:
            _lib.raise_for_status(self.lib.vipmi_trim(self.handle), "vipmi_trim")
        finally:
            self._in_call.release()
        return True

    def destroy(self):
        """Synchronise the context's stream and free every workspace it owns (hipFree); the object is dead afterwards."""
        h, self.handle = self.handle, None
        if h:
            self.lib.vipmi_destroy(h)

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


_gate_box = {"gate": None}
_tls = threading.local()                 # asynchronous mode is a property of the calling THREAD (see set_async)


def _mode():
    return getattr(_tls, "mode", (False, 0))


def _gate():
    """Process-wide gate shared by every context that runs in asynchronous mode (include/vipmi.h: vipmi_gate)."""
    with _ctx_lock:
        if _gate_box["gate"] is None:
            lib = _lib.load()
            h = ctypes.c_void_p()
            _lib.raise_for_status(lib.vipmi_gate_create(ctypes.byref(h)), "vipmi_gate_create")
            _gate_box["gate"] = h
        return _gate_box["gate"]


def set_async(on=True, reserve_cus=None):
    """Asynchronous (pipelined) mode for the calls of THIS THREAD: they enqueue all work on the current torch stream and
    never synchronise (the eigensolver's convergence check is latched on the device and read by ``check_deferred()``).
    Each (device, stream) pair gets its own vipmi_ctx / workspace, so independent calls issued on two streams overlap:
    the latency-bound eigensolver of one cube (16 workgroups) runs beside the FFT derotation of the previous one
    (``reserve_cus`` CUs can be kept free for it; measured best: 0).  All asynchronous contexts share a gate that runs
    the chip-filling half of the calls one at a time in issue order (otherwise identical calls drift into lock step and
    the chip idles while every stream sits in its eigensolver).

    The mode is thread-local and reaches a context when that thread next calls into it (``Context.call``), so a
    pipelined region on one thread (pca_many, the annular / 4-D fronts) neither switches a concurrent caller on another
    thread to deferred error checks nor has the mode switched off under it."""
    if reserve_cus is None:          # default 0: since the shear kernels take their work from dynamic queues, a CU that is
        reserve_cus = int(os.environ.get("VIPMI_RESERVE_CUS", "0"))   # busy with the other call's eigensolver costs nothing
    _tls.mode = (bool(on), int(reserve_cus) if on else 0)


def is_async():
    return _mode()[0]


_side_streams = {}


def side_streams(depth, device=None):
    """``depth`` cached torch streams of the device (independent sub-problems of ONE call -- spectral channels --
    are issued round-robin on them in asynchronous mode, see psfsub/pca_fullfr.py)."""
    torch = require_gpu()
    dev = torch.cuda.current_device() if device is None else int(device)
    lst = _side_streams.setdefault(dev, [])
    while len(lst) < depth:
        lst.append(torch.cuda.Stream(device=dev))
    return lst[:depth]


def check_deferred():
    """Synchronise every context this thread has called in asynchronous mode since its last check and raise if a
    deferred error (eigensolver non-convergence) was latched on one of them."""
    touched = getattr(_tls, "touched", None) or {}
    _tls.touched = {}
    first = None
    for c in touched.values():
        if not c.handle:
            continue
        with c._in_call:
            st = c.lib.vipmi_check_deferred(c.handle)
        if st != 0 and first is None:
            try:
                _lib.raise_for_status(st, "vipmi_check_deferred")
            except Exception as e:              # keep checking (and thereby clearing) the other contexts first
                first = e
    if first is not None:
        raise first


def all_contexts():
    """Every live context (one per (device, stream) that has run a call)."""
    with _ctx_lock:
        return list(_ctx_cache.values())


def get_context(device=None):
    torch = require_gpu()
    dev = torch.cuda.current_device() if device is None else int(device)
    key = (dev, int(torch.cuda.current_stream(dev).cuda_stream))
    with _ctx_lock:
        c = _ctx_cache.get(key)
        if c is None:
            if len(_ctx_cache) >= max(1, MAX_CONTEXTS):
                for old_key, old in list(_ctx_cache.items()):    # least recently used first
                    if old.trim():
                        del _ctx_cache[old_key]
                        break
            c = Context(dev)
            _ctx_cache[key] = c
        else:
            _ctx_cache.move_to_end(key)
        return c


def release_workspaces():
    """Empty the context cache: every idle context is trimmed (stream synchronised, all of its hipMalloc'ed workspaces
    freed) and dropped; handles that other code still holds stay valid and are destroyed with their last reference.
    The next call on a stream builds a fresh context; use between phases of a long-running process that worked on
    large cubes."""
    with _ctx_lock:
        for key, c in list(_ctx_cache.items()):
            if c.trim():
                del _ctx_cache[key]


# ---- array plumbing ------------------------------------------------------------------------------

def is_device_tensor(x):
    try:
        import torch
    except ImportError:
        return False
    return isinstance(x, torch.Tensor)


def to_device_f32(x, device=None):
    """numpy / torch -> contiguous float32 cuda tensor (no copy when already so)."""
    torch = require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
    if isinstance(x, torch.Tensor):
        t = x.to(device=dev, dtype=torch.float32)
    else:
        a = np.ascontiguousarray(x)
        if a.dtype in (np.float64, np.float16, np.int32, np.int64, np.int16, np.uint8):
            # convert on the device: a float64 cube of C2 size costs 60 ms in numpy's astype, 15 ms to upload as is
            # (same round-to-nearest-even conversion)
            t = torch.from_numpy(a).to(dev).to(torch.float32)
        else:
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    return t.contiguous()


# ---- results back to numpy --------------------------------------------------------------------------------------------------
# A cube-sized result (full_output: recon, residuals, derotated residuals -- 419 MB each at C2) copied with t.cpu() lands in freshly
# mapped pageable memory: the runtime stages it through a pinned bounce buffer and the host copy faults every page in, 25-44 ms per
# array (10-17 GB/s) on a 6 ms computation.  Big results therefore go straight into PINNED host memory from torch's caching host
# allocator and are handed to the caller as numpy arrays that own their block: one DMA at link speed (~56 GB/s), and once the caller
# drops an array its block goes back to the allocator's pool and serves the next call already pinned and mapped (the usual loop --
# `out = pca(..., full_output=True)` per iteration -- reuses the same few blocks).  A caller that keeps every result keeps the blocks
# too: beyond VIPMI_PINNED_OUT_MB (default 16384) of blocks handed out and still alive, results fall back to pageable memory.
_PIN_MIN_BYTES = 512 << 10      # (a 1 MB final frame already gains 0.14 ms over a pageable destination: tools/frame_d2h_probe.py)
_pin_out = {"bytes": 0}
_pin_lock = threading.Lock()


class _PinnedBlock:
    """Owner of one pinned host block: numpy arrays made from it keep it (and the torch tensor whose storage the block is) alive."""

    def __init__(self, h):
        self.h = h
        self.__array_interface__ = h.numpy().__array_interface__


def _np_to_torch_dtype(dt):
    torch = _torch()
    return {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}.get(np.dtype(dt))


def to_host_many(tensors, dtypes=None):
    """cuda tensors -> numpy arrays (see above); ``dtypes``: one numpy dtype (or None = the tensor's) per tensor, the conversion
    float32 <-> float64 done on the device before the copy.  All copies are enqueued before the one synchronisation."""
    import weakref
    torch = _torch()
    dtypes = list(dtypes) if dtypes is not None else [None] * len(tensors)
    cap = int(os.environ.get("VIPMI_PINNED_OUT_MB", "16384")) << 20
    outs, pend = [], []
    for t, dt in zip(tensors, dtypes):
        want = _np_to_torch_dtype(dt) if dt is not None else None
        if want is not None and t.dtype != want and t.dtype.is_floating_point:
            t = t.to(want)                         # (on the device: numpy's astype of a cube costs more than its copy)
        nbytes = t.numel() * t.element_size()
        h = None
        if nbytes >= _PIN_MIN_BYTES and cap > 0:
            with _pin_lock:
                ok = _pin_out["bytes"] + nbytes <= cap
                if ok:
                    _pin_out["bytes"] += nbytes
            if ok:
                try:
                    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                except RuntimeError:
                    h = None
                if h is None:
                    with _pin_lock:
                        _pin_out["bytes"] -= nbytes
        if h is None:
            a = t.cpu().numpy()
        else:
            h.copy_(t.contiguous(), non_blocking=True)
            pend.append(h)
            blk = _PinnedBlock(h)
            a = np.asarray(blk)                    # (a.base is blk, and so is the base of every view the caller takes of a)

            def _release(nb=nbytes):
                with _pin_lock:
                    _pin_out["bytes"] -= nb
            weakref.finalize(blk, _release)        # (blk -- and with it the pinned tensor -- lives as long as those arrays)
        if dt is not None and a.dtype != np.dtype(dt):
            a = a.astype(dt, copy=False)
        outs.append(a)
    if pend:
        torch.cuda.current_stream().synchronize()
    return outs


def to_host(t, dtype=None):
    return to_host_many([t], [dtype])[0]


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def empty(shape, dtype=None, device=None):
    torch = require_gpu()
    return torch.empty(shape, dtype=dtype or torch.float32,
                       device=torch.device("cuda", torch.cuda.current_device() if device is None else int(device)))


def host_f64(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return a, a.ctypes.data_as(ctypes.c_void_p)


# ---- thin wrappers (device tensors in, device tensors out) ----------------------------------------

def scale(M, mode, out=None):
    ctx = get_context(M.device.index)
    out = empty(M.shape, device=M.device.index) if out is None else out
    n, P = M.shape
    ctx.call("vipmi_scale_f32", ptr(M), ptr(out), n, P, SCALE_MODES[mode])
    return out


def apply_mask(M, mask_u8, fill=0.0, out=None):
    ctx = get_context(M.device.index)
    out = empty(M.shape, device=M.device.index) if out is None else out
    P = mask_u8.numel()
    n = M.numel() // P
    ctx.call("vipmi_apply_mask_f32", ptr(M), ptr(out), n, P, ptr(mask_u8), ctypes.c_float(fill))
    return out


def lincomb(x, y=None, a=1.0, b=1.0):
    """a*x + b*y (y optional) elementwise on float32 cuda tensors of equal shape."""
    ctx = get_context(x.device.index)
    x = x.contiguous()
    yy = None if y is None else y.contiguous()
    out = empty(tuple(x.shape), device=x.device.index)
    ctx.call("vipmi_lincomb_f32", ptr(x), ptr(yy), ctypes.c_float(a), ctypes.c_float(b), x.numel(), ptr(out))
    return out


def gram_batched(M3):
    """M3: (batch, n, P) float32 cuda tensor -> (batch, n, n) float64 Gram matrices, one launch."""
    torch = _torch()
    M3 = M3.contiguous()
    ctx = get_context(M3.device.index)
    b, n, P = M3.shape
    G = empty((b, n, n), torch.float64, M3.device.index)
    ctx.call("vipmi_gram_batched_f32", ptr(M3), b, n, P, ptr(G))
    return G


def gram(M):
    torch = _torch()
    ctx = get_context(M.device.index)
    n, P = M.shape
    G = empty((n, n), torch.float64, M.device.index)
    ctx.call("vipmi_gram_f32", ptr(M), n, P, M.stride(0), ptr(G))
    return G


def cross_gram(A, B):
    torch = _torch()
    ctx = get_context(A.device.index)
    na, P = A.shape
    nb = B.shape[0]
    C = empty((na, nb), torch.float64, A.device.index)
    ctx.call("vipmi_cross_gram_f32", ptr(A), na, ptr(B), nb, P, A.stride(0), ptr(C))
    return C


def eigh(G):
    """G: (batch, n, n) or (n, n) float64 cuda tensor (destroyed).  Returns (evals desc, evecs rows)."""
    torch = _torch()
    ctx = get_context(G.device.index)
    single = G.dim() == 2
    Gb = G.unsqueeze(0) if single else G
    batch, n, _ = Gb.shape
    Gb = Gb.contiguous()
    evals = empty((batch, n), torch.float64, G.device.index)
    evecs = empty((batch, n, n), torch.float64, G.device.index)
    ctx.call("vipmi_eigh_f64", ptr(Gb), batch, n, ptr(evals), ptr(evecs))
    return (evals[0], evecs[0]) if single else (evals, evecs)


def eigh_topk(G, k, nact=None, all_evals=False):
    """Leading k eigenpairs of G: (batch, n, n) or (n, n) float64 cuda tensor (destroyed).  Returns
    (evals (.., k) descending -- all n of them with ``all_evals`` --, evecs (.., k, n) rows).  nact: optional int32
    cuda tensor of active sizes (not with ``all_evals``)."""
    torch = _torch()
    ctx = get_context(G.device.index)
    single = G.dim() == 2
    Gb = (G.unsqueeze(0) if single else G).contiguous()
    batch, n, _ = Gb.shape
    evals = torch.zeros((batch, n), dtype=torch.float64, device=G.device)
    evecs = torch.zeros((batch, n, n), dtype=torch.float64, device=G.device)
    if all_evals:
        if n > MAX_EIGH_LDS_N:
            _warn_slow_eigh(n, "the whole spectrum was requested")
        ctx.call("vipmi_eigh_spectrum_f64", ptr(Gb), batch, n, int(k), ptr(evals), ptr(evecs))
        ev, ec = evals, evecs[:, :k, :]
    else:
        ctx.call("vipmi_eigh_topk_f64", ptr(Gb), batch, n, int(k), ptr(nact), ptr(evals), ptr(evecs))
        if n > MAX_EIGH_LDS_N and int(ctx.get_option("eigh_fast_last_reason")) != 0:
            _warn_slow_eigh(n, "the verified subspace iteration gave up (reason %d)" % int(ctx.get_option("eigh_fast_last_reason")))
        ev, ec = evals[:, :k], evecs[:, :k, :]
    return (ev[0], ec[0]) if single else (ev, ec)


def _warn_slow_eigh(n, why):
    import warnings
    warnings.warn("vip_amd: %d frames (more than %d): %s -- solved by the exact tridiagonal path with its vectors in global memory "
                  "(slow: seconds, not milliseconds)" % (n, MAX_EIGH_LDS_N, why), RuntimeWarning, stacklevel=3)


def topk_native(n, k):
    """True when the leading-k tridiagonal solvers serve (n, k): up to 512 frames and 64 vectors in LDS
    (eigh_tri.hip), 128..16384 frames and any number of vectors with the matrix in L2 / HBM (eigh_tri_large.hip)."""
    return 0 < k <= n and ((n <= 512 and k <= 64) or 128 <= n <= MAX_EIGH_N)


MAX_EIGH_LDS_N = 6144    # the leading-k solver keeps three vectors of n doubles in LDS: matrices up to 6144 x 6144 that way
MAX_EIGH_N = 16384       # beyond (round 6): the same solver with those vectors in global memory -- slow, exact, any request
                         # (leading pairs when the verified fast path gives up, the whole spectrum, CEVR); 2.1 GB matrix at 16384


def eigh_topk_fast(G, k):
    """Leading k eigenpairs of ONE positive semi-definite (n, n) float64 cuda tensor through the verified fast path alone
    (csrc/eigh_chfsi.hip).  Returns (evals (k,) descending, evecs (k, n) rows) or None when it does not converge within its
    budget / the sizes are outside its range; G is not modified."""
    torch = _torch()
    ctx = get_context(G.device.index)
    n = G.shape[0]
    G = G.contiguous()
    evals = torch.zeros((n,), dtype=torch.float64, device=G.device)
    evecs = torch.zeros((int(k), n), dtype=torch.float64, device=G.device)
    conv = ctypes.c_int(0)
    ctx.call("vipmi_eigh_topk_fast_f64", ptr(G), n, int(k), ptr(evals), ptr(evecs), ctypes.byref(conv))
    return (evals[:k], evecs) if conv.value else None


def eigh_beyond_lds(G, k=None):
    """Leading pairs of a Gram matrix of more than MAX_EIGH_N = 16384 frames (up to there the exact hand-written solvers serve every
    request, eigh_tri_large.hip): only the verified Chebyshev subspace iteration (csrc/eigh_chfsi.hip) is tried, for the leading
    ``k`` pairs; returns (evals (k,), evecs (k, n)).  There is NO library fallback: a request for the whole spectrum (``k`` None),
    a size outside the fast path's range or a fast path that does not converge raises NotImplementedError."""
    torch = _torch()
    n = int(G.shape[0])
    if k is None:
        raise NotImplementedError("the whole spectrum of a cube / library of %d frames (more than %d) is not available on the "
                                  "device eigensolvers; ask for the leading components only" % (n, MAX_EIGH_N))
    fast = None
    try:
        fast = eigh_topk_fast(G.to(torch.float64), k)
    except ValueError:
        pass                                       # (sizes outside the fast path's range)
    if fast is None:
        ctx = get_context(G.device.index)
        raise NotImplementedError("the leading %d eigenpairs of a %d-frame Gram matrix (more than %d frames) did not converge in the "
                                  "verified subspace iteration (reason %d); no other solver serves this size" % (
                                      int(k), n, MAX_EIGH_N, int(ctx.get_option("eigh_fast_last_reason"))))
    return fast


def _pca_project_large(M, k, ref, want_recon, want_pcs, want_evals):
    """``pca_project`` for more than MAX_EIGH_N reference frames: Gram and both projection products on the device
    kernels, the eigendecomposition through ``eigh_beyond_lds``."""
    torch = _torch()
    ctx = get_context(M.device.index)
    n, P = M.shape
    dev = M.device.index
    refm = M if ref is None else ref
    nref = refm.shape[0]
    w, E = eigh_beyond_lds(gram(refm), None if want_evals else k)
    ev = w[:k]
    keep = (ev > ev[0] * 1e-12)
    Ek = (E[:k] * keep[:, None]).to(torch.float32).contiguous()       # (k, nref)
    inv = torch.where(keep, 1.0 / torch.sqrt(torch.clamp(ev, min=1e-300)), torch.zeros_like(ev)).to(torch.float32).contiguous()
    res = empty((n, P), device=dev)
    if ref is None:
        T = empty((k, P), device=dev)
        ctx.call("vipmi_rowspace_gemm_f32", ptr(Ek), ptr(M), k, n, P, None, ptr(T))
        C = Ek.t().contiguous()
        ctx.call("vipmi_subtract_gemm_f32", ptr(M), ptr(C), ptr(T), n, k, P, ptr(res), None)
        pcs = (T * inv[:, None]).contiguous() if want_pcs else None
    else:
        V = empty((k, P), device=dev)                                   # V = S^-1 E^T ref  (orthonormal rows)
        ctx.call("vipmi_rowspace_gemm_f32", ptr(Ek), ptr(refm), k, nref, P, ptr(inv), ptr(V))
        C64 = empty((n, k), torch.float64, dev)
        ctx.call("vipmi_cross_gram_f32", ptr(M), n, ptr(V), k, P, P, ptr(C64))
        C = C64.to(torch.float32).contiguous()
        ctx.call("vipmi_subtract_gemm_f32", ptr(M), ptr(C), ptr(V), n, k, P, ptr(res), None)
        pcs = V if want_pcs else None
    recon = lincomb(M, res, 1.0, -1.0) if want_recon else None
    return res, recon, pcs, (w if want_evals else None)


def pca_project(M, k, ref=None, want_recon=False, want_pcs=False, want_evals=False):
    """residuals (and optionally recon, pcs, evals) of M w.r.t. the top-k PCs of ref (default M)."""
    torch = _torch()
    ctx = get_context(M.device.index)
    n, P = M.shape
    refm = M if ref is None else ref
    nref = refm.shape[0]
    if nref > MAX_EIGH_N:
        return _pca_project_large(M, k, ref, want_recon, want_pcs, want_evals)
    dev = M.device.index
    res = empty((n, P), device=dev)
    recon = empty((n, P), device=dev) if want_recon else None
    pcs = empty((k, P), device=dev) if want_pcs else None
    evals = empty((nref,), torch.float64, dev) if want_evals else None
    ctx.call("vipmi_pca_project_f32", ptr(M), n, ptr(refm), nref, P, k, ptr(res), ptr(recon), ptr(pcs), ptr(evals))
    return res, recon, pcs, evals


INTERP_MODES = {"nearneig": 0, "bilinear": 1, "bicubic": 2, "lanczos4": 3}
BORDER_MODES = {"constant": 0, "edge": 1, "symmetric": 2, "reflect": 3, "wrap": 4}     # derotation.py:294-305
import contextvars

# rotation used by derotate(): set by rotation_mode() / with_rotation; a ContextVar, so concurrent calls from different
# threads (or asyncio tasks) with different `imlib` values never see each other's choice
_ROTATION = contextvars.ContextVar("vipmi_rotation", default=("vip-fft", "lanczos4", "constant", None))


def other_mask_value(mask_val):
    """frame_rotate's ``mask_val`` when it is neither NaN nor 0 (the two cases every device entry takes as flags):
    the float the 'vip-fft' rotation resets matching pixels to (derotation.py:133-140,324-326); else None."""
    if mask_val is None:
        return None
    try:
        v = float(mask_val)
    except (TypeError, ValueError):
        raise TypeError("mask_val must be a number")
    return None if (v != v or v == 0.0) else v


def check_border(border_mode):
    if border_mode not in BORDER_MODES:
        raise ValueError("Opencv `border_mode` not recognized.")
    return border_mode


def check_imlib(imlib, interpolation="lanczos4"):
    """Validate the reference's (imlib, interpolation) switches: 'vip-fft' (parity path) or 'opencv' (interpolating
    rotation, derotation.py:279-305); the others are not implemented on the device."""
    imlib = str(getattr(imlib, "value", imlib))
    interpolation = str(getattr(interpolation, "value", interpolation))
    if imlib == "vip-fft":
        return imlib, interpolation
    if imlib == "opencv":
        if interpolation not in INTERP_MODES:
            raise ValueError("Opencv interpolation method `%s` is not recognized" % interpolation)
        return imlib, interpolation
    if imlib in ("skimage", "torch-fft", "ndimage"):
        raise NotImplementedError("vip_amd implements imlib='vip-fft' and imlib='opencv' only (got %r)" % imlib)
    raise ValueError("Image transformation library not recognized")


class rotation_mode:
    """``with rotation_mode(imlib, interpolation):`` -- every ``derotate`` inside uses that rotation."""

    def __init__(self, imlib, interpolation="lanczos4", border_mode="constant", mask_val=None):
        self.mode = list(check_imlib(imlib, interpolation))
        self.mode.append(check_border(border_mode) if self.mode[0] == "opencv" else "constant")
        self.mode.append(other_mask_value(mask_val))

    def __enter__(self):
        self.token = _ROTATION.set(tuple(self.mode))
        return self

    def __exit__(self, *exc):
        _ROTATION.reset(self.token)
        return False


def with_rotation(fn):
    """Decorator: run ``fn`` under the rotation named by its ``imlib`` / ``interpolation`` arguments."""
    import functools
    import inspect
    sig = inspect.signature(fn)

    @functools.wraps(fn)
    def wrapper(*a, **k):
        b = sig.bind(*a, **k)
        b.apply_defaults()
        extra = b.arguments.get("rot_options", None) or {}
        with rotation_mode(b.arguments.get("imlib", "vip-fft"), b.arguments.get("interpolation", "lanczos4"),
                           extra.get("border_mode", "constant"), extra.get("mask_val")):
            return fn(*a, **k)
    return wrapper


def rotate_interp(cube, angles, interpolation="lanczos4", cxy=None, out=None, border_mode="constant"):
    """frames rotated by -angles with OpenCV's warpAffine arithmetic (imlib='opencv'); centre = frame_center."""
    ctx = get_context(cube.device.index)
    n, Ny, Nx = cube.shape
    if Ny != Nx:
        raise ValueError("derotation on the device requires square frames")
    out = empty(cube.shape, device=cube.device.index) if out is None else out
    cx, cy = (float(Nx // 2), float(Ny // 2)) if cxy is None else (float(cxy[0]), float(cxy[1]))   # coords.py:61-100
    ah, ap = host_f64(angles)
    ctx.call("vipmi_rotate_interp_f32", ptr(cube), ap, n, Ny, cx, cy, INTERP_MODES[interpolation],
             BORDER_MODES[check_border(border_mode)], ptr(out))
    return out


def derotate(cube, angles, mask_nan=True, mask_zero=False, method="auto", out=None, mask_val=None):
    """``mask_nan`` / ``mask_zero``: the NaN and 0 cases of frame_rotate's ``mask_val``; any other value comes in
    through ``mask_val`` or the enclosing ``rotation_mode`` (rot_options['mask_val'] of the front-ends)."""
    rot = _ROTATION.get()
    if rot[0] == "opencv":
        return rotate_interp(cube, angles, rot[1], out=out, border_mode=rot[2])
    ctx = get_context(cube.device.index)
    n, Ny, Nx = cube.shape
    if Ny != Nx:
        raise ValueError("vip-fft derotation on the device requires square frames")
    out = empty(cube.shape, device=cube.device.index) if out is None else out
    ah, ap = host_f64(angles)
    mv = other_mask_value(mask_val)
    if mv is None and mask_val is None:
        mv = rot[3]
    if mv is not None:
        # (the reference compares its float32 frames with the Python float in float32 -- numpy's scalar promotion --, so a
        # mask_val such as 0.1 matches the float32 pixels that hold float32(0.1): the kernel's comparison exactly)
        ctx.call("vipmi_derotate_maskval_f32", ptr(cube), ap, n, Ny, ptr(out), ctypes.c_float(mv), ROT_METHODS[method])
        return out
    ctx.call("vipmi_derotate_f32", ptr(cube), ap, n, Ny, ptr(out), int(bool(mask_nan)), int(bool(mask_zero)),
             ROT_METHODS[method])
    return out


def collapse(cube, mode="median", w=None, trim_n=50):     # n=50: subsampling.py:30 default of cube_collapse
    ctx = get_context(cube.device.index)
    n = cube.shape[0]
    P = cube[0].numel()
    out = empty(cube.shape[1:], device=cube.device.index)
    wt = to_device_f32(w, cube.device.index) if w is not None else None
    ctx.call("vipmi_collapse_f32", ptr(cube), n, P, COLLAPSE_MODES[mode], ptr(wt), int(trim_n), ptr(out))
    return out


def project_batched(M, E):
    """R[b] = M[b] - E[b]^T (E[b] M[b]):  M (nb, n, P) float32, E (nb, k, n) float32 rows = eigenvectors."""
    ctx = get_context(M.device.index)
    nb, n, P = M.shape
    k = E.shape[1]
    R = empty((nb, n, P), device=M.device.index)
    ctx.call("vipmi_project_batched_f32", ptr(M), ptr(E), nb, n, k, P, ptr(R))
    return R


def collapse_batched(cubes, mode="median", w=None, trim_n=50):
    """collapse of every cube of a contiguous stack (batch, n, ...) -> (batch, ...) in one launch."""
    ctx = get_context(cubes.device.index)
    if not cubes.is_contiguous():
        cubes = cubes.contiguous()
    nb, n = cubes.shape[0], cubes.shape[1]
    P = cubes[0, 0].numel()
    out = empty((nb,) + tuple(cubes.shape[2:]), device=cubes.device.index)
    wt = to_device_f32(w, cubes.device.index) if w is not None else None
    ctx.call("vipmi_collapse_batched_f32", ptr(cubes), nb, n, P, COLLAPSE_MODES[mode], ptr(wt), int(trim_n), ptr(out))
    return out


def pca_fullframe_f64(cube64, angles, ncomp, scaling=None, mask_u8=None, collapse_mode="median", full_output=False):
    """Fused 3-D ADI path for a FLOAT64 cuda cube (vipmi_pca_fullframe_f64: the per-pixel temporal mean carried in float64, the
    float32 kernels on what is left).  ``scaling``: None or any of the reference's modes.  Returns the frame (float32 cuda
    tensor), or (frame, pcs, recon, residuals, residuals_der) with ``full_output``."""
    torch = _torch()
    assert cube64.dtype == torch.float64 and cube64.is_cuda
    cube64 = cube64.contiguous()
    ctx = get_context(cube64.device.index)
    n, N, _ = cube64.shape
    dev = cube64.device.index
    frame = empty((N, N), device=dev)
    pcs = recon = res = der = None
    if full_output:
        k = min(int(ncomp), n)
        pcs = empty((k, N, N), device=dev)
        recon = empty((n, N, N), device=dev)
        res = empty((n, N, N), device=dev)
        der = empty((n, N, N), device=dev)
    ah, ap = host_f64(angles)
    ctx.call("vipmi_pca_fullframe_f64", ptr(cube64), ap, n, N, int(ncomp), SCALE_MODES[scaling], ptr(mask_u8),
             COLLAPSE_MODES[collapse_mode], ptr(frame), ptr(pcs), ptr(recon), ptr(res), ptr(der))
    return (frame, pcs, recon, res, der) if full_output else frame


def pca_fullframe_hostin(cube_np, angles, ncomp, mask_u8=None, collapse_mode="median", full_output=False, device=None):
    """Fused 3-D ADI path for a float32 numpy cube still in host memory (vipmi_pca_fullframe_hostin_f32: the library uploads it in
    blocks of 64 frames and forms the Gram matrix under the copy).  Returns like ``pca_fullframe``; the full_output tuple ends with
    nothing extra -- the uploaded cube is dropped."""
    torch = require_gpu()
    assert isinstance(cube_np, np.ndarray) and cube_np.dtype == np.float32 and cube_np.flags.c_contiguous and cube_np.ndim == 3
    dev = torch.cuda.current_device() if device is None else int(device)
    ctx = get_context(dev)
    n, N, _ = cube_np.shape
    k = min(int(ncomp), n)
    cube = empty((n, N, N), device=dev)
    frame = empty((N, N), device=dev)
    pcs = recon = res = der = None
    if full_output:
        pcs = empty((k, N, N), device=dev)
        recon = empty((n, N, N), device=dev)
        res = empty((n, N, N), device=dev)
        der = empty((n, N, N), device=dev)
    ah, ap = host_f64(angles)
    ctx.call("vipmi_pca_fullframe_hostin_f32", cube_np.ctypes.data_as(ctypes.c_void_p), ptr(cube), ap, n, N, int(ncomp), ptr(mask_u8),
             COLLAPSE_MODES[collapse_mode], ptr(frame), ptr(pcs), ptr(recon), ptr(res), ptr(der))
    if full_output:
        return frame, pcs, recon, res, der
    return frame


def pca_fullframe(cube, angles, ncomp, scaling=None, mask_u8=None, collapse_mode="median",
                  full_output=False):
    """Fused 3-D ADI path.  Returns frame or (frame, pcs, recon, residuals, residuals_der)."""
    ctx = get_context(cube.device.index)
    n, N, _ = cube.shape
    dev = cube.device.index
    k = min(int(ncomp), n)
    frame = empty((N, N), device=dev)
    pcs = recon = res = der = None
    if full_output:
        pcs = empty((k, N, N), device=dev)
        recon = empty((n, N, N), device=dev)
        res = empty((n, N, N), device=dev)
        der = empty((n, N, N), device=dev)
    ah, ap = host_f64(angles)
    ctx.call("vipmi_pca_fullframe_f32", ptr(cube), ap, n, N, int(ncomp), SCALE_MODES[scaling], ptr(mask_u8),
             COLLAPSE_MODES[collapse_mode], ptr(frame), ptr(pcs), ptr(recon), ptr(res), ptr(der))
    if full_output:
        return frame, pcs, recon, res, der
    return frame
