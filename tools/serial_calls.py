"""Ten un-pipelined pca() calls on a resident cube (for rocprofv3 --kernel-trace: gaps between the kernels of a call)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi_device
n, N, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (400, 512, 20)
from vip_amd import backend as _B
for _o in sys.argv[4:]:
    _a, _b = _o.split("="); _B.get_context().set_option(_a, int(_b))
ct, ang = synth_adi_device(n, N, seed=0)
for _ in range(10):
    fr = pca(ct, ang, ncomp=k, verbose=False, check_memory=False).cpu()
torch.cuda.synchronize()
