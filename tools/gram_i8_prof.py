"""A few int8 Gram calls for rocprofv3 (--kernel-trace / --pmc): python tools/gram_i8_prof.py [n N]  (gram_i8 mode 1 only)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi_device
n, N = (int(a) for a in sys.argv[1:3]) if len(sys.argv) > 2 else (400, 512)
ct, ang = synth_adi_device(n, N, seed=0)
M = ct.reshape(n, -1)
ctx = B.get_context()
ctx.set_option("gram_i8", 1)
for _ in range(4):
    B.gram(M)
torch.cuda.synchronize()
