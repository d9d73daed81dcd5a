"""200 un-pipelined pca() calls on a 50 x 128 x 128 cube (for rocprofv3 --hip-trace --stats: host time per HIP API)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi_device
ct, ang = synth_adi_device(50, 128, seed=0)
for _ in range(200):
    fr = pca(ct, ang, ncomp=5, verbose=False, check_memory=False).cpu()
torch.cuda.synchronize()
