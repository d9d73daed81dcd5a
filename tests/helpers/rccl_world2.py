"""Launched by tests/test_gpu_pca.py under torch.distributed.run with two ranks on two GPUs (RCCL): every sharded routine
of vip_amd.dist -- the torch.distributed partitions and the C entry vipmi_pca_fullframe_sharded_f32 on an RCCL
communicator created by the library -- against the single-GPU calls on the same cube; rank 0 prints max|diff| per case."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("nccl", rank=rank, world_size=world)
assert dist.get_world_size() == world and world >= 2

from vip_amd import dist as D
from vip_amd.psfsub import pca, pca_annular
from vip_amd.synth import synth_adi


def report(name, got, ref):
    got = got.cpu().numpy() if hasattr(got, "cpu") else np.asarray(got)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), name
    if rank == 0:
        print("case %s maxdiff %.3e" % (name, np.nanmax(np.abs(got - ref))))


for n, N, k, collapse in ((24, 64, 4, "median"), (31, 96, 5, "mean"), (17, 45, 3, "median")):      # ragged frame / row splits
    cube, ang = synth_adi(n, N, seed=n)
    ref = pca(cube, ang, ncomp=k, collapse=collapse, verbose=False)
    cube_t = torch.from_numpy(cube).cuda()
    report("single_cube_%d_%d" % (n, N), D.pca_single_cube(cube_t, ang, k, collapse=collapse), ref)
    comm = D.RcclComm()
    assert (comm.rank, comm.world) == (rank, world)
    report("c_entry_%d_%d" % (n, N), D.pca_single_cube_rccl(cube_t, ang, k, comm, collapse=collapse), ref)
    comm.destroy()
cube, ang = synth_adi(30, 128, seed=3)
cube_t = torch.from_numpy(cube).cuda()
report("annular", D.pca_annular(cube_t, ang, ncomp=3, asize=16, fwhm=4),
       pca_annular(cube, ang, ncomp=3, asize=16, fwhm=4, verbose=False))
c4 = np.stack([cube, cube[::-1] * 0.5, cube * 0.25])
report("ifs_4d", D.pca_4d(torch.from_numpy(c4).cuda(), ang, ncomp=4, verbose=False, check_memory=False)[0],
       pca(c4, ang, ncomp=4, verbose=False))
dist.barrier()
if rank == 0:
    print("OK")
dist.destroy_process_group()
