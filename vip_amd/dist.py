"""Multi-GPU sharding of the PSF-subtraction path: one process per GPU
(`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The path partitions into independent units -- cubes (survey mode / contrast-curve loops), IFS channels
of a 4-D cube, annuli of an annular PCA -- so units are dealt to ranks and there is NO collective inside
the data path; the only communication is the final gather of the small per-unit products
(frames of N*N floats, or annulus residual columns) to rank 0.

Every function takes the per-unit compute callable as an argument (default: the device implementation
of `vip_amd.psfsub`), which is what lets the world_size-2 gloo tests exercise the sharding / gather
logic on CPU with a numpy stand-in for the device kernels.
"""
import numpy as np


def _dist():
    import torch.distributed as dist
    return dist


def world_info():
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_round_robin(n_items, rank=None, world=None):
    """Indices of the units owned by ``rank`` (unit i -> rank i % world): 39 channels over 8 ranks give
    5/5/5/5/5/5/5/4."""
    r, w = world_info()
    rank = r if rank is None else rank
    world = w if world is None else world
    return list(range(rank, n_items, world))


def shard_balanced(weights, rank=None, world=None):
    """Longest-processing-time assignment of weighted units (annuli weighted by pixel count: the outer
    annulus of C3 is 15x the innermost) -> sorted list of unit indices owned by ``rank``.  Deterministic,
    identical on every rank."""
    r, w = world_info()
    rank = r if rank is None else rank
    world = w if world is None else world
    order = sorted(range(len(weights)), key=lambda i: (-float(weights[i]), i))
    load = [0.0] * world
    owner = [0] * len(weights)
    for i in order:
        j = min(range(world), key=lambda q: (load[q], q))
        owner[i] = j
        load[j] += float(weights[i])
    return sorted(i for i in range(len(weights)) if owner[i] == rank)


def _work_device(*arrays):
    """Device on which a sharded routine assembles its buffers: that of the first cuda tensor among ``arrays``, else
    the communication device (cpu under gloo)."""
    import torch
    for a in arrays:
        if isinstance(a, torch.Tensor) and a.is_cuda:
            return a.device
    return _comm_device()


def _is_cuda(a):
    try:
        import torch
    except ImportError:
        return False
    return isinstance(a, torch.Tensor) and a.is_cuda


def _comm_device():
    import torch
    dist = _dist()
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _all_gather_ragged(mine, counts, dev):
    """all_gather of per-rank stacks whose leading sizes ``counts[r]`` differ (known on every rank): RCCL takes ragged
    lists as they are; gloo needs equal shapes, so the stacks are padded to the largest."""
    import torch
    dist = _dist()
    rank, world = world_info()
    tail = tuple(mine.shape[1:])
    if world == 1:
        return [mine]
    cdev = _comm_device()
    if mine.device != cdev:                       # gloo with device tensors (single-GPU test rig): stage through the host
        return [p.to(mine.device) for p in _all_gather_ragged(mine.to(cdev), counts, cdev)]
    if dist.get_backend() == "nccl":
        out = [torch.empty((c,) + tail, dtype=mine.dtype, device=dev) for c in counts]
        dist.all_gather(out, mine.contiguous())
        return out
    cmax = max(counts)
    pad = torch.zeros((cmax,) + tail, dtype=mine.dtype, device=dev)
    pad[:mine.shape[0]] = mine
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return [bufs[r][:counts[r]] for r in range(world)]


def gather_units(local, n_units, unit_shape, owners, dtype=None):
    """All ranks contribute their units {index: array}; every rank returns the full (n_units, *unit_shape) array.
    ``owners[r]`` = sorted unit indices of rank r (None: round-robin).  One all_gather of the per-rank stacks -- every
    unit crosses the links once (no reduction over zero-filled copies of the whole stack)."""
    import torch
    rank, world = world_info()
    dev = _work_device(*local.values())
    dtype = dtype or torch.float32
    if owners is None:
        owners = [shard_round_robin(n_units, r, world) for r in range(world)]
    mine_idx = owners[rank]
    if sorted(local.keys()) != list(mine_idx):
        raise ValueError("gather_units: this rank computed units %r but owns %r" % (sorted(local.keys()), list(mine_idx)))
    mine = torch.zeros((len(mine_idx),) + tuple(unit_shape), dtype=dtype, device=dev)
    for j, i in enumerate(mine_idx):
        a = local[i]
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        mine[j] = t.to(device=dev, dtype=dtype)
    parts = _all_gather_ragged(mine, [len(o) for o in owners], dev)
    buf = torch.zeros((n_units,) + tuple(unit_shape), dtype=dtype, device=dev)
    for r in range(world):
        if len(owners[r]):
            buf[torch.as_tensor(list(owners[r]), dtype=torch.long, device=dev)] = parts[r]
    return buf


def pca_cubes(cubes, angle_lists, compute=None, **kwargs):
    """Survey mode: a list of cubes (same frame size), one full-frame PCA each, cubes dealt round-robin.
    Returns the stack of final frames (n_cubes, N, N) on every rank."""
    if compute is None:
        from .psfsub import pca as compute
    mine = shard_round_robin(len(cubes))
    local = {}
    for i in mine:
        local[i] = compute(cubes[i], angle_lists[i], **kwargs)
    shape = tuple(cubes[0].shape[-2:])
    return gather_units(local, len(cubes), shape, None)


def pca_4d(cube4d, angle_list, ncomp=1, collapse_ifs="mean", compute=None, collapse=None, **kwargs):
    """4-D cube without ``scale_list`` (reference psfsub/pca_fullfr.py:544-658): channels dealt round-robin,
    per-channel ADI frames gathered, spectral collapse on the gathered stack.  Returns (frame, ifs_adi_frames)."""
    default_compute = compute is None
    if compute is None:
        from .psfsub import pca as compute
    if collapse is None:
        from .preproc import cube_collapse as collapse
    nch = cube4d.shape[0]
    ncomps = ncomp if isinstance(ncomp, list) else [ncomp] * nch
    mine = shard_round_robin(nch)
    local = {}
    if (default_compute and len(mine) > 1 and _is_cuda(cube4d) and isinstance(ncomp, (int, np.integer))
            and not (set(kwargs) - {"verbose", "check_memory", "nproc"})
            and 0 < int(ncomp) <= min(64, cube4d.shape[1]) and cube4d.shape[1] <= 512):
        # plain ADI with one integer ncomp: the rank's channels through the batched stages of the 4-D front (ONE Gram,
        # eigensolver, projection, derotation and collapse launch for all of them: csrc/api.hip vipmi_pca_4d_f32's
        # building blocks) instead of a pca() call per channel -- the same arithmetic, 2.5x faster at C4 on one GPU
        import torch
        from .psfsub.pca_fullfr import _adi_pca_channels_batched
        from .preproc.parangles import check_pa_vector
        sub = cube4d[torch.as_tensor(mine, device=cube4d.device)].to(torch.float32).contiguous()
        frames = _adi_pca_channels_batched(sub, check_pa_vector(np.asarray(angle_list, dtype=np.float64)), int(ncomp),
                                           None, None, "median", None, True)
        for j, ch in enumerate(mine):
            local[ch] = frames[j]
    else:
        for ch in mine:
            local[ch] = compute(cube4d[ch], angle_list, ncomp=ncomps[ch], **kwargs)
    _mark("per-channel ADI PCA (own channels)")
    ifs = gather_units(local, nch, tuple(cube4d.shape[-2:]), None)
    _mark("all_gather (per-channel frames)")
    ifs_np = ifs.cpu().numpy()
    frame = collapse(ifs_np, mode=collapse_ifs)
    _mark("spectral collapse (host)")
    return frame, ifs_np


def _segment_owners(plan):
    """owner rank of every segment of an annulus plan (balanced by pixel count, identical on every rank)."""
    rank, world = world_info()
    weights = [len(s["pix"]) for s in plan]
    owner = [0] * len(plan)
    for r in range(world):
        for si in shard_balanced(weights, r, world):
            owner[si] = r
    return owner


def pca_annular_residuals(cube, angle_list, plan, residual_fn):
    """Annuli of an annular PCA dealt over ranks by pixel count.  ``plan`` = list of segment dicts
    (vip_amd.psfsub.pca_local.annulus_plan); ``residual_fn(seg) -> (n, npx) residuals`` computes one
    segment.  Returns cube_out (n, y, x) on every rank (what ``full_output`` needs); segments are applied in plan order
    so the 1-pixel overlap of the last annulus is resolved exactly as in the reference (pca_local.py:786-787).
    One all_gather of every rank's residual columns.  For the final frame alone use ``pca_annular_frame``, which never
    assembles the whole residual cube on any rank."""
    import torch
    rank, world = world_info()
    owner = _segment_owners(plan)
    n = cube.shape[0]
    y, x = cube.shape[-2:]
    dev = _work_device(cube)
    cols = []
    for si, seg in enumerate(plan):
        if owner[si] == rank:
            r = residual_fn(seg)
            r = r if isinstance(r, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(r))
            cols.append(r.to(device=dev, dtype=torch.float32)[:, :len(seg["pix"])].t())       # (npx, n): pixel-major
    mine = torch.cat(cols, dim=0).contiguous() if cols else torch.zeros((0, n), dtype=torch.float32, device=dev)
    counts = [sum(len(seg["pix"]) for si, seg in enumerate(plan) if owner[si] == r) for r in range(world)]
    parts = _all_gather_ragged(mine, counts, dev)
    out = torch.zeros((n, y * x), dtype=torch.float32, device=dev)
    offs = [0] * world
    for si, seg in enumerate(plan):
        r = owner[si]
        npx = len(seg["pix"])
        pix = torch.from_numpy(np.asarray(seg["pix"], dtype=np.int64)).to(dev)
        out[:, pix] = parts[r][offs[r]:offs[r] + npx].t()
        offs[r] += npx
    return out.reshape(n, y, x)


# ---- one cube sharded over the GPUs ("C2/C5 single cube") --------------------------------------------

def _split(total, world):
    """Contiguous, near-equal blocks: [start, stop) of every rank."""
    base, rem = divmod(total, world)
    edges = [0]
    for r in range(world):
        edges.append(edges[-1] + base + (1 if r < rem else 0))
    return [(edges[r], edges[r + 1]) for r in range(world)]


# ---- per-phase wall times of the sharded paths (bench.py's strong-scaling legs: what did the collectives cost?) ------------------
import threading as _threading

_phase_tls = _threading.local()       # the clock belongs to the THREAD that switched it on: sharded routines called from several
                                      # host threads (each rank is one process, but a rank may run them from worker threads) neither
                                      # see nor disturb each other's phases


def _phase_state():
    st = getattr(_phase_tls, "st", None)
    if st is None:
        st = _phase_tls.st = {"on": False, "acc": {}, "last": 0.0}
    return st


def phase_timing(on=True):
    """Switch the phase clock of the sharded routines on (clears it) or off (returns {phase: milliseconds}) for the calling
    thread.  With the clock on, every phase boundary synchronises the device, so the phases do not overlap: run it on a step of
    its own."""
    import time
    _phases = _phase_state()
    if on:
        _phases.update(on=True, acc={}, last=time.perf_counter())
        return None
    _phases["on"] = False
    return {k: 1e3 * v for k, v in _phases["acc"].items()}


def _mark(name):
    _phases = _phase_state()
    if not _phases["on"]:
        return
    import time
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except ImportError:
        pass
    now = time.perf_counter()
    _phases["acc"][name] = _phases["acc"].get(name, 0.0) + now - _phases["last"]
    _phases["last"] = now


def _all_to_all(send_chunks, recv_shapes, dtype, dev, out=None):
    """send_chunks[r] goes to rank r; returns the list of chunks received (recv_shapes[r] from rank r).
    RCCL: one all_to_all over the xGMI links; gloo (CPU tests) has no all_to_all, so pairwise isend/irecv.
    out: optional list of preallocated CONTIGUOUS receive buffers (views of the tensor the caller assembles the chunks in:
    saves the concatenation pass).  World 1: nothing moves -- the chunk itself is returned (a view; round 3 copied it and the
    caller concatenated it again: 2 x 7 ms of the 174 ms of a C5 call on one GPU)."""
    import torch
    dist = _dist()
    rank, world = world_info()
    if world == 1:
        return [send_chunks[0]]
    cdev = _comm_device()
    if torch.device(dev) != cdev:                 # gloo with device tensors: stage through the host
        got = _all_to_all([c.to(cdev) for c in send_chunks], recv_shapes, dtype, cdev)
        if out is not None:
            for o, g in zip(out, got):
                o.copy_(g)
            return out
        return [g.to(dev) for g in got]
    recv = out if out is not None else [torch.empty(s, dtype=dtype, device=dev) for s in recv_shapes]
    send = [c.contiguous() for c in send_chunks]
    if dist.get_backend() == "nccl":
        dist.all_to_all(recv, send)
        return recv
    recv[rank].copy_(send[rank])
    reqs = []
    for r in range(world):
        if r != rank:
            reqs.append(dist.isend(send[r], dst=r))
            reqs.append(dist.irecv(recv[r], src=r))
    for q in reqs:
        q.wait()
    return recv


class DeviceOps:
    """The per-rank compute of ``pca_single_cube`` on the MI355X (float32 cuda tensors in and out)."""

    def to_dev(self, a):
        from . import backend as B
        return B.to_device_f32(a)

    def gram(self, M):
        from . import backend as B
        return B.gram(M)

    def leading(self, G, k):
        from . import backend as B
        ev, ec = B.eigh_topk(G, k)
        return ev, ec

    def residuals(self, M, ev, ec):
        """M - E E^T M for the slab M (n x P_g); E rows = ec (k x n)."""
        from . import backend as B
        torch = B._torch()
        keep = (ev > ev[0] * 1e-12).to(torch.float32)
        E = (ec.to(torch.float32) * keep[:, None]).contiguous()        # (k, n)
        n, P = M.shape
        k = E.shape[0]
        ctx = B.get_context(M.device.index)
        T = B.empty((k, P), device=M.device.index)
        ctx.call("vipmi_rowspace_gemm_f32", B.ptr(E), B.ptr(M), k, n, P, None, B.ptr(T))
        R = B.empty((n, P), device=M.device.index)
        C = E.t().contiguous()                                          # (n, k)
        ctx.call("vipmi_subtract_gemm_f32", B.ptr(M), B.ptr(C), B.ptr(T), n, k, P, B.ptr(R), None)
        return R

    def derotate(self, frames, angles, mask_zero=False):
        from . import backend as B
        return B.derotate(frames.contiguous(), angles, mask_nan=not mask_zero, mask_zero=mask_zero)

    def segment_residuals(self, cube_t, seg):
        """(n, npx) residuals of one annulus segment (plain ADI annular PCA: gather the segment's pixel columns, one
        batched launch for the n per-frame library decompositions; reference psfsub/pca_local.py:708-757,830-909)."""
        from . import backend as B
        from .psfsub.pca_local import _pack_libs
        torch = B._torch()
        n = cube_t.shape[0]
        P = cube_t[0].numel()
        dev = cube_t.device
        pix_h = np.asarray(seg["pix"], dtype=np.int32)
        npx0 = pix_h.size
        if npx0 % 4:                                 # zero columns: 16-byte aligned rows, Gram unchanged
            pix_h = np.concatenate([pix_h, np.full(4 - npx0 % 4, -1, dtype=np.int32)])
        pix = torch.from_numpy(pix_h).to(dev)
        npx = int(pix.numel())
        ctx = B.get_context(dev.index)
        A = B.empty((n, npx), device=dev.index)
        ctx.call("vipmi_gather_f32", B.ptr(cube_t), n, P, B.ptr(pix), npx, B.ptr(A))
        idx, ln, max_lib = _pack_libs(seg["libs"])
        idx_t, ln_t = torch.from_numpy(idx).to(dev), torch.from_numpy(ln).to(dev)
        R = B.empty((n, npx), device=dev.index)
        ctx.call("vipmi_annular_residuals_f32", B.ptr(A), n, npx, B.ptr(idx_t), B.ptr(ln_t), max_lib, int(seg["ncomp"]),
                 B.ptr(R))
        return R[:, :npx0]

    def collapse(self, cube, mode):
        """cube: (n, P_g, 1)-shaped view -> (P_g,)"""
        from . import backend as B
        return B.collapse(cube, mode).reshape(-1)


def pca_single_cube(cube, angle_list, ncomp, collapse="median", ops=None):
    """Full-frame ADI PCA of ONE cube sharded over all ranks (every rank passes the same ``cube`` / ``angle_list``;
    only its own slab is touched before the exchanges):

      pixels (rows of the frames) sharded -> partial Gram -> all_reduce(n x n float64) -> identical leading
      eigenvectors on every rank -> local project/subtract -> all_to_all (pixel slabs -> whole frames, frames sharded)
      -> local derotation -> all_to_all back (-> pixel slabs of all frames) -> local collapse -> all_gather(frame).

    Returns the final frame (y, x) on every rank.  The data-path collectives are exactly those of SURVEY 8(e)."""
    import torch
    dist = _dist()
    rank, world = world_info()
    ops = ops or DeviceOps()
    n, y, x = cube.shape
    angle_list = np.asarray(angle_list, dtype=np.float64)
    if angle_list.shape[0] != n:
        raise ValueError("`angle_list` vector has wrong length. It must equal the number of frames in the cube")
    if ncomp <= 0:
        raise ValueError("Number of PCs too low. It should be > 0.")
    k = min(int(ncomp), n)
    rows = _split(y, world)            # pixel rows owned by every rank
    frs = _split(n, world)             # frames owned by every rank (after the first exchange)
    y0, y1 = rows[rank]
    f0, f1 = frs[rank]
    # 1. own pixel slab, partial Gram, all-reduce
    M = ops.to_dev(np.ascontiguousarray(cube[:, y0:y1, :]).reshape(n, -1) if isinstance(cube, np.ndarray)
                   else cube[:, y0:y1, :].reshape(n, -1))
    G = ops.gram(M)
    dev = G.device
    _mark("gram (own pixel slab)")
    if world > 1:
        if G.device != _comm_device():            # gloo with device tensors: stage through the host
            Gh = G.to(_comm_device())
            dist.all_reduce(Gh, op=dist.ReduceOp.SUM)
            G = Gh.to(dev)
        else:
            dist.all_reduce(G, op=dist.ReduceOp.SUM)
    _mark("all_reduce (n x n float64 Gram)")
    # 2. identical decomposition everywhere, local residual slab
    ev, ec = ops.leading(G, k)
    _mark("eigensolver (replicated)")
    R = ops.residuals(M, ev, ec)                                               # (n, (y1-y0)*x)
    _mark("project / subtract (own slab)")
    R3 = R.reshape(n, y1 - y0, x)
    # 3. slabs -> whole frames
    send = [R3[a:b] for (a, b) in frs]
    recv_shapes = [(f1 - f0, r1 - r0, x) for (r0, r1) in rows]
    parts = _all_to_all(send, recv_shapes, R.dtype, dev)
    frames = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)          # (f1-f0, y, x)
    _mark("all_to_all 1 (pixel slabs -> whole frames)")
    return _derotate_exchange_collapse(frames, angle_list, frs, rows, collapse, ops, dev)


class RcclComm:
    """An RCCL communicator owned by libvipmi (include/vipmi.h: vipmi_rccl_*), one rank per process / GPU.  The 128-byte
    unique id is made on rank 0 and handed to the other ranks through the already initialised ``torch.distributed``
    group (any backend: it is host data); with world 1 nothing is exchanged."""

    def __init__(self, device=None):
        import ctypes
        from . import _lib, backend as B
        self._lib = _lib
        self.ctx = B.get_context(device)         # (imports torch first: its HIP runtime must be the one in the process)
        lib = _lib.load()
        rank, world = world_info()
        self.rank, self.world = rank, world
        ident = (ctypes.c_char * 128)()
        if rank == 0:
            _lib.raise_for_status(lib.vipmi_rccl_unique_id(ident), "vipmi_rccl_unique_id")
        if world > 1:
            box = [bytes(ident)]
            _dist().broadcast_object_list(box, src=0)
            ident = (ctypes.c_char * 128).from_buffer_copy(box[0])
        handle = ctypes.c_void_p()
        _lib.raise_for_status(lib.vipmi_rccl_comm_create(self.ctx.handle, ident, rank, world, ctypes.byref(handle)),
                              "vipmi_rccl_comm_create")
        self.handle = handle

    def destroy(self):
        h, self.handle = self.handle, None
        if h:
            self._lib.load().vipmi_rccl_comm_destroy(h)


def pca_single_cube_rccl(cube, angle_list, ncomp, comm, collapse="median"):
    """``pca_single_cube`` through the C entry ``vipmi_pca_fullframe_sharded_f32``: the same partition (pixel rows for the
    decomposition, frames for the derotation), with the collectives issued by the library itself on an RCCL communicator
    (``RcclComm``) instead of ``torch.distributed``.  Every rank passes the same ``cube`` (only its row slab is
    touched) or, as a cuda tensor of shape (n, y1 - y0, x), its slab; returns the final frame on every rank.
    The C entry has no ``scaling`` / ``mask_center_px`` arguments (include/vipmi.h) and stops at 6144 frames."""
    from . import backend as B
    torch = B._torch()
    n = cube.shape[0]
    angle_list = np.ascontiguousarray(angle_list, dtype=np.float64)
    if angle_list.shape[0] != n:
        raise ValueError("`angle_list` vector has wrong length. It must equal the number of frames in the cube")
    N = cube.shape[2]
    y0, y1 = _split(N, comm.world)[comm.rank]
    if cube.shape[1] == N:
        slab = B.to_device_f32(np.ascontiguousarray(cube[:, y0:y1, :]) if isinstance(cube, np.ndarray)
                               else cube[:, y0:y1, :].contiguous())
    elif cube.shape[1] == y1 - y0:
        slab = B.to_device_f32(cube)
    else:
        raise TypeError("cube must be the whole (n, N, N) cube or this rank's (n, rows, N) slab")
    mode = B.COLLAPSE_MODES[collapse]
    frame = torch.empty((N, N), dtype=torch.float32, device=slab.device)
    comm.ctx.call("vipmi_pca_fullframe_sharded_f32", comm.handle, comm.rank, comm.world, B.ptr(slab),
                  angle_list.ctypes.data, n, N, int(ncomp), int(mode), B.ptr(frame))
    return frame


def _derotate_exchange_collapse(frames, angle_list, frs, rows, collapse, ops, dev, **rot):
    """Tail shared by the sharded single-cube and annular paths: this rank holds the whole residual frames of its
    frame shard -> local derotation -> all_to_all (whole frames -> pixel-row slabs of ALL frames) -> local collapse of
    the slab -> all_gather of the final frame's row slabs."""
    import torch
    dist = _dist()
    rank, world = world_info()
    n = frs[-1][1]
    x = frames.shape[-1]
    y0, y1 = rows[rank]
    f0, f1 = frs[rank]
    # 4. derotate own frames
    der = ops.derotate(frames, angle_list[f0:f1], **rot) if f1 > f0 else frames
    _mark("derotation (own frames)")
    # 5. whole frames -> slabs of all frames
    send = [der[:, r0:r1, :] for (r0, r1) in rows]
    recv_shapes = [(b - a, y1 - y0, x) for (a, b) in frs]
    # (the chunk of rank r holds its frames [a, b) of this rank's rows: contiguous pieces of the slab -- received in place)
    if world == 1:
        slab = der.reshape(n, -1, 1)
    else:
        slab3 = torch.empty((n, y1 - y0, x), dtype=der.dtype, device=der.device)
        _all_to_all(send, recv_shapes, der.dtype, dev, out=[slab3[a:b] for (a, b) in frs])
        slab = slab3.reshape(n, -1, 1)                                         # (n, P_g, 1)
    _mark("all_to_all 2 (whole frames -> pixel slabs)")
    # 6. collapse own pixels, gather the frame
    mine = ops.collapse(slab, collapse).reshape(y1 - y0, x)
    _mark("collapse (own pixel slab)")
    if world == 1:
        return mine
    pieces = _all_gather_ragged(mine.contiguous(), [r1 - r0 for (r0, r1) in rows], dev)
    out = torch.cat(pieces, dim=0)
    _mark("all_gather (final frame)")
    return out


def pca_annular_frame(cube, angle_list, plan, residual_fn, collapse="median", ops=None, mask_zero=False):
    """Final frame of an annular ADI PCA with the partition of SURVEY 8(e), row "C3 annular":

      annuli (segments) dealt to the ranks by pixel count -> each rank computes the residual COLUMNS (n frames x its
      pixels) of its segments -> all_to_all (pixel columns -> whole frames, frames sharded; every residual crosses the
      links once) -> sharded derotation -> all_to_all back (whole frames -> pixel-row slabs of all frames) -> sharded
      collapse -> all_gather of the frame's row slabs.

    No rank ever holds the whole residual cube, no reduction over zero-padded copies, and derotation + collapse -- a
    seventh of the single-GPU time at C3 -- scale with the number of ranks too.  ``plan`` / ``residual_fn`` as in
    ``pca_annular_residuals``; segments are applied in plan order (1-pixel overlap of the last annulus)."""
    import torch
    rank, world = world_info()
    ops = ops or DeviceOps()
    n = cube.shape[0]
    y, x = cube.shape[-2:]
    angle_list = np.asarray(angle_list, dtype=np.float64)
    owner = _segment_owners(plan)
    dev = _work_device(cube)
    frs = _split(n, world)
    rows = _split(y, world)
    f0, f1 = frs[rank]
    cols = []
    for si, seg in enumerate(plan):
        if owner[si] == rank:
            r = residual_fn(seg)
            r = r if isinstance(r, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(r))
            cols.append(r.to(device=dev, dtype=torch.float32)[:, :len(seg["pix"])])
    mine = torch.cat(cols, dim=1) if cols else torch.zeros((n, 0), dtype=torch.float32, device=dev)   # (n, npx_mine)
    counts = [sum(len(seg["pix"]) for si, seg in enumerate(plan) if owner[si] == r) for r in range(world)]
    _mark("annular residuals (own segments)")
    # pixel columns -> whole frames of the own frame shard
    send = [mine[a:b] for (a, b) in frs]
    recv_shapes = [(f1 - f0, counts[r]) for r in range(world)]
    parts = _all_to_all(send, recv_shapes, torch.float32, dev)
    frames = torch.zeros((f1 - f0, y * x), dtype=torch.float32, device=dev)
    offs = [0] * world
    for si, seg in enumerate(plan):
        r = owner[si]
        npx = len(seg["pix"])
        pix = torch.from_numpy(np.asarray(seg["pix"], dtype=np.int64)).to(dev)
        frames[:, pix] = parts[r][:, offs[r]:offs[r] + npx]
        offs[r] += npx
    _mark("all_to_all 1 (pixel columns -> whole frames)")
    rot = {"mask_zero": True} if mask_zero else {}
    return _derotate_exchange_collapse(frames.reshape(f1 - f0, y, x), angle_list, frs, rows, collapse, ops, dev, **rot)


def pca_annular(cube, angle_list, ncomp=1, asize=4, fwhm=4, radius_int=0, n_segments=1, delta_rot=(0.1, 1),
                min_frames_lib=2, max_frames_lib=200, collapse="median", theta_init=0, ops=None):
    """``vip_amd.psfsub.pca_annular(cube, angle_list, ...)`` (plain ADI: no reference cube, no scaling) with the annuli
    sharded over the ranks -- BASELINE.json configs[2].  Every rank passes the same cube; returns the final frame on
    every rank."""
    from .psfsub.pca_local import cached_annulus_plan
    from .preproc.parangles import check_pa_vector
    rank, world = world_info()
    if world == 1 and ops is None and _is_cuda(cube):
        # one rank: nothing to shard -- the single-GPU front (all annuli through ONE batched eigensolve) is the
        # same arithmetic and 1.4x faster than the per-segment path below
        from .psfsub import pca_annular as _pca_annular
        return _pca_annular(cube, angle_list, ncomp=ncomp, asize=asize, fwhm=fwhm, radius_int=radius_int,
                            n_segments=n_segments, delta_rot=delta_rot, min_frames_lib=min_frames_lib,
                            max_frames_lib=max_frames_lib, collapse=collapse, theta_init=theta_init, verbose=False)
    ops = ops or DeviceOps()
    angle_list = check_pa_vector(np.asarray(angle_list, dtype=np.float64))
    n, y, x = cube.shape
    plan, _ = cached_annulus_plan((y, x), angle_list, radius_int, fwhm, asize, n_segments, delta_rot, ncomp,
                                  min_frames_lib, max_frames_lib, theta_init)
    cube_dev = ops.to_dev(cube)
    return pca_annular_frame(cube_dev, angle_list, plan, lambda seg: ops.segment_residuals(cube_dev, seg), collapse=collapse,
                             ops=ops, mask_zero=bool(radius_int))
