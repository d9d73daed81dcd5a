"""A few median collapses of a C2-sized cube for a counter / trace run: python tools/prof_median.py [real|gauss] [n N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vip_amd import backend as B
kind = sys.argv[1] if len(sys.argv) > 1 else "gauss"
n, N = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (400, 512)
if kind == "real":
    from vip_amd.psfsub import pca
    from vip_amd.synth import synth_adi_device
    ct, ang = synth_adi_device(n, N, seed=0)
    out = pca(ct, ang, ncomp=20, full_output=True, verbose=False, check_memory=False)
    cube = [o for o in out if torch.is_tensor(o) and o.ndim == 3 and o.shape[0] == n][-1]
else:
    cube = torch.randn(n, N, N, device="cuda")
for _ in range(5):
    B.collapse(cube, "median")
torch.cuda.synchronize()
