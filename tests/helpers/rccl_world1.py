"""Run by tests/test_gpu_pca.py in a subprocess (its own timeout): the C entry vipmi_pca_fullframe_sharded_f32 on a
world-1 RCCL communicator against pca() on the same cube; prints max|diff| per case."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from vip_amd import dist as D
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi

comm = D.RcclComm()
assert (comm.rank, comm.world) == (0, 1)
for n, N, k, collapse in ((24, 64, 4, "median"), (30, 96, 5, "mean"), (17, 45, 3, "median")):
    cube, ang = synth_adi(n, N, seed=n)
    got = D.pca_single_cube_rccl(cube, ang, k, comm, collapse=collapse).cpu().numpy()
    ref = pca(cube, ang, ncomp=k, collapse=collapse, verbose=False)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    print("case %d %d %d %s maxdiff %.3e" % (n, N, k, collapse, np.nanmax(np.abs(got - ref))))
comm.destroy()
print("OK")
