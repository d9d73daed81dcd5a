"""CPU, world_size 2, gloo: the sharding / gather logic of vip_amd.dist with the oracle as the per-unit
compute stand-in (the device kernels need a GPU; the distributed plumbing does not)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_cpu as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vip_amd import dist as D
        from vip_amd.psfsub.pca_local import annulus_plan
        res = {}
        # --- sharding helpers
        res["rr"] = D.shard_round_robin(39)
        res["bal"] = D.shard_balanced([3205, 9644, 16064, 22516, 28940, 35408, 41804, 48028])
        # --- survey mode: 5 cubes over 2 ranks
        cubes, angs = zip(*[O.synth_adi(8, 24, seed=s) for s in range(5)])
        calls = []

        def comp(c, a, **kw):
            calls.append(1)
            return O.pca_fullframe(c, a, **kw)
        frames = D.pca_cubes(list(cubes), list(angs), compute=comp, ncomp=2)
        res["frames"] = frames.numpy()
        res["ncalls"] = len(calls)
        # --- 4-D: 3 channels over 2 ranks
        c4 = np.stack([O.synth_adi(8, 24, seed=10 + i)[0] for i in range(3)])
        a4 = np.linspace(0, 70, 8)
        frame, ifs = D.pca_4d(c4, a4, ncomp=2, compute=lambda c, a, ncomp: O.pca_fullframe(c, a, ncomp=ncomp),
                              collapse=O.cube_collapse)
        res["frame4d"], res["ifs"] = frame, ifs
        # --- annular: segments over 2 ranks
        cube, ang = O.synth_adi(12, 32, seed=3)
        angc = O.check_pa_vector(ang)
        plan = annulus_plan((32, 32), angc, 0, 4, 5, 1, (0.1, 1), 2, 2, 200)

        def resid(seg):
            yy, xx = np.divmod(seg["pix"], 32)
            m = cube[:, yy, xx]
            out = np.zeros_like(m)
            for fr in range(12):
                lib = m[seg["libs"][fr]]
                V = O.svd_wrapper(lib, "lapack", min(seg["ncomp"], min(lib.shape)))
                out[fr] = m[fr] - (m[fr] @ V.T) @ V
            return out
        res["cube_out"] = D.pca_annular_residuals(cube, angc, plan, resid).numpy()
        # --- ONE cube sharded over the ranks: all_reduce of the Gram + two all_to_all exchanges (ragged splits:
        #     11 frames / 23 rows over 2 ranks), numpy stand-ins for the device kernels
        class NumpyOps:
            def to_dev(self, a):
                return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))

            def gram(self, M):
                m = M.numpy().astype(np.float64)
                return torch.from_numpy(m @ m.T)

            def leading(self, G, k):
                w, v = np.linalg.eigh(G.numpy())
                return torch.from_numpy(w[::-1][:k].copy()), torch.from_numpy(v[:, ::-1][:, :k].T.copy())

            def residuals(self, M, ev, ec):
                m = M.numpy().astype(np.float64)
                E = ec.numpy()
                return torch.from_numpy((m - E.T @ (E @ m)).astype(np.float32))

            def derotate(self, frames, angles, mask_zero=False):
                return torch.from_numpy(O.cube_derotate(frames.numpy(), np.asarray(angles),
                                                        mask_val=0 if mask_zero else np.nan))

            def collapse(self, cube, mode):
                return torch.from_numpy(O.cube_collapse(cube.numpy(), mode)).reshape(-1)
        cs, as_ = O.synth_adi(11, 23, seed=21)
        res["single"] = D.pca_single_cube(cs, as_, 3, ops=NumpyOps()).numpy()
        # the phase clock bench.py's strong-scaling legs report (one extra step, synchronised at the phase boundaries)
        D.phase_timing(True)
        again = D.pca_single_cube(cs, as_, 3, ops=NumpyOps()).numpy()
        res["phases"] = D.phase_timing(False)
        res["phases_same"] = bool(np.array_equal(again, res["single"], equal_nan=True))
        # --- annular PCA with the SURVEY 8(e) partition: residual columns -> all_to_all to frame shards -> sharded
        #     derotation -> all_to_all back -> sharded collapse -> gather (ragged: 12 frames / 32 rows / 3 segments)
        res["ann_frame"] = D.pca_annular_frame(cube, angc, plan, resid, collapse="median", ops=NumpyOps()).numpy()
        plan_ri = annulus_plan((32, 32), angc, 4, 4, 5, 2, (0.1, 1), 2, 2, 200)      # inner hole + 2 segments per annulus
        res["ann_frame_ri"] = D.pca_annular_frame(cube, angc, plan_ri, resid, collapse="mean", ops=NumpyOps(),
                                                  mask_zero=True).numpy()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_world_gloo(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ranks = [out[r] for r in range(world)]
    r0, r1 = ranks[0], ranks[1]
    # disjoint cover
    assert sorted(sum((r["rr"] for r in ranks), [])) == list(range(39)) and r0["rr"][:3] == [0, world, 2 * world]
    assert sorted(sum((r["bal"] for r in ranks), [])) == list(range(8))
    w = np.array([3205, 9644, 16064, 22516, 28940, 35408, 41804, 48028])
    loads = [w[r["bal"]].sum() for r in ranks]
    assert max(loads) - min(loads) <= w.max() * 0.2
    # each rank computed only its share, all hold the full result, identical to the serial computation
    assert [r["ncalls"] for r in ranks] == [len(range(q, 5, world)) for q in range(world)]
    cubes, angs = zip(*[O.synth_adi(8, 24, seed=s) for s in range(5)])
    serial = np.stack([O.pca_fullframe(c, a, ncomp=2) for c, a in zip(cubes, angs)])
    for r in ranks:
        assert np.array_equal(r["frames"], serial)
    cs, as_ = O.synth_adi(11, 23, seed=21)
    ref_single = O.pca_fullframe(cs, as_, ncomp=3)
    for r in ranks:
        assert r["single"].shape == (23, 23)
        assert np.nanmax(np.abs(r["single"] - ref_single)) < 2e-5
    assert np.array_equal(r0["single"], r1["single"], equal_nan=True)
    for r in (r0, r1):
        assert r["phases_same"]
        assert {"all_reduce (n x n float64 Gram)", "all_to_all 1 (pixel slabs -> whole frames)", "derotation (own frames)",
                "all_to_all 2 (whole frames -> pixel slabs)", "all_gather (final frame)"} <= set(r["phases"])
        assert all(v >= 0 for v in r["phases"].values())
    c4 = np.stack([O.synth_adi(8, 24, seed=10 + i)[0] for i in range(3)])
    a4 = np.linspace(0, 70, 8)
    f4 = O.pca_4d(c4, a4, ncomp=2, full_output=True)
    for r in ranks:
        assert np.allclose(r["frame4d"], f4[0], atol=1e-6)
        assert np.allclose(r["ifs"], f4[5], atol=1e-6)
    cube, ang = O.synth_adi(12, 32, seed=3)
    co = O.pca_annular(cube, ang, asize=5, ncomp=2, fwhm=4, delta_rot=(0.1, 1), full_output=True)[0]
    fr_ref = O.pca_annular(cube, ang, asize=5, ncomp=2, fwhm=4, delta_rot=(0.1, 1))
    fr_ri = O.pca_annular(cube, ang, asize=5, ncomp=2, fwhm=4, delta_rot=(0.1, 1), radius_int=4, n_segments=2,
                          collapse="mean")
    for r in ranks:
        assert np.abs(r["cube_out"] - co).max() < 1e-5
        assert np.nanmax(np.abs(r["ann_frame"] - fr_ref)) < 2e-5
        assert np.array_equal(np.isnan(r["ann_frame"]), np.isnan(fr_ref))
        assert np.nanmax(np.abs(r["ann_frame_ri"] - fr_ri)) < 2e-5
    assert all(np.array_equal(r["ann_frame"], r0["ann_frame"], equal_nan=True) for r in ranks)
