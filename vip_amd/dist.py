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


def _comm_device():
    import torch
    dist = _dist()
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_units(local, n_units, unit_shape, owners, dtype=None):
    """All ranks contribute their units {index: array}; every rank returns the full (n_units, *unit_shape)
    array (all_gather of a zero-filled stack + ownership mask: units are disjoint, so a sum is exact)."""
    import torch
    dist = _dist()
    rank, world = world_info()
    dev = _comm_device()
    dtype = dtype or torch.float32
    buf = torch.zeros((n_units,) + tuple(unit_shape), dtype=dtype, device=dev)
    for i, a in local.items():
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        buf[i] = t.to(device=dev, dtype=dtype)
    if world > 1:
        # disjoint ownership: a sum over ranks reassembles the stack exactly (x + 0 == x)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
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
    if compute is None:
        from .psfsub import pca as compute
    if collapse is None:
        from .preproc import cube_collapse as collapse
    nch = cube4d.shape[0]
    ncomps = ncomp if isinstance(ncomp, list) else [ncomp] * nch
    mine = shard_round_robin(nch)
    local = {}
    for ch in mine:
        local[ch] = compute(cube4d[ch], angle_list, ncomp=ncomps[ch], **kwargs)
    ifs = gather_units(local, nch, tuple(cube4d.shape[-2:]), None)
    ifs_np = ifs.cpu().numpy()
    frame = collapse(ifs_np, mode=collapse_ifs)
    return frame, ifs_np


def pca_annular_residuals(cube, angle_list, plan, residual_fn):
    """Annuli of an annular PCA dealt over ranks by pixel count.  ``plan`` = list of segment dicts
    (vip_amd.psfsub.pca_local.annulus_plan); ``residual_fn(seg) -> (n, npx) residuals`` computes one
    segment.  Returns cube_out (n, y, x) on every rank; segments are applied in plan order so the
    1-pixel overlap of the last annulus is resolved exactly as in the reference (pca_local.py:786-787)."""
    import torch
    dist = _dist()
    rank, world = world_info()
    weights = [len(s["pix"]) for s in plan]
    mine = set(shard_balanced(weights))
    n = cube.shape[0]
    y, x = cube.shape[-2:]
    dev = _comm_device()
    out = torch.zeros((n, y * x), dtype=torch.float32, device=dev)
    for si, seg in enumerate(plan):
        owner_has = si in mine
        npx = len(seg["pix"])
        buf = torch.zeros((n, npx), dtype=torch.float32, device=dev)
        if owner_has:
            r = residual_fn(seg)
            r = r if isinstance(r, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(r))
            buf.copy_(r.to(device=dev, dtype=torch.float32)[:, :npx])
        if world > 1:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        pix = torch.from_numpy(np.asarray(seg["pix"], dtype=np.int64)).to(dev)
        out[:, pix] = buf
    return out.reshape(n, y, x)


# ---- one cube sharded over the GPUs ("C2/C5 single cube") --------------------------------------------

def _split(total, world):
    """Contiguous, near-equal blocks: [start, stop) of every rank."""
    base, rem = divmod(total, world)
    edges = [0]
    for r in range(world):
        edges.append(edges[-1] + base + (1 if r < rem else 0))
    return [(edges[r], edges[r + 1]) for r in range(world)]


def _all_to_all(send_chunks, recv_shapes, dtype, dev):
    """send_chunks[r] goes to rank r; returns the list of chunks received (recv_shapes[r] from rank r).
    RCCL: one all_to_all over the xGMI links; gloo (CPU tests) has no all_to_all, so pairwise isend/irecv."""
    import torch
    dist = _dist()
    rank, world = world_info()
    recv = [torch.empty(s, dtype=dtype, device=dev) for s in recv_shapes]
    send = [c.contiguous() for c in send_chunks]
    if world == 1:
        recv[0].copy_(send[0])
        return recv
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

    def derotate(self, frames, angles):
        from . import backend as B
        return B.derotate(frames, angles)

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
    if world > 1:
        dist.all_reduce(G, op=dist.ReduceOp.SUM)
    # 2. identical decomposition everywhere, local residual slab
    ev, ec = ops.leading(G, k)
    R = ops.residuals(M, ev, ec)                                               # (n, (y1-y0)*x)
    R3 = R.reshape(n, y1 - y0, x)
    # 3. slabs -> whole frames
    send = [R3[a:b] for (a, b) in frs]
    recv_shapes = [(f1 - f0, r1 - r0, x) for (r0, r1) in rows]
    parts = _all_to_all(send, recv_shapes, R.dtype, dev)
    frames = torch.cat(parts, dim=1)                                           # (f1-f0, y, x)
    # 4. derotate own frames
    der = ops.derotate(frames, angle_list[f0:f1]) if f1 > f0 else frames
    # 5. whole frames -> slabs of all frames
    send = [der[:, r0:r1, :] for (r0, r1) in rows]
    recv_shapes = [(b - a, y1 - y0, x) for (a, b) in frs]
    parts = _all_to_all(send, recv_shapes, der.dtype, dev)
    slab = torch.cat(parts, dim=0).reshape(n, -1, 1)                           # (n, P_g, 1)
    # 6. collapse own pixels, gather the frame
    mine = ops.collapse(slab, collapse).reshape(y1 - y0, x)
    if world == 1:
        return mine
    pieces = [torch.empty((r1 - r0, x), dtype=mine.dtype, device=dev) for (r0, r1) in rows]
    if dist.get_backend() == "nccl" or all(p.shape == pieces[0].shape for p in pieces):
        dist.all_gather(pieces, mine.contiguous())
    else:                                   # gloo all_gather needs equal shapes: pad to the largest slab
        hmax = max(r1 - r0 for (r0, r1) in rows)
        pad = torch.zeros((hmax, x), dtype=mine.dtype, device=dev)
        pad[:y1 - y0] = mine
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad)
        pieces = [bufs[r][:rows[r][1] - rows[r][0]] for r in range(world)]
    return torch.cat(pieces, dim=0)
