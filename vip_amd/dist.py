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
