import os, sys
sys.path.insert(0, "/root/repo")
import torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi_device
def t(fn, reps=9):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
ctx = B.get_context()
ct, ang = synth_adi_device(400, 512, seed=0)
out = pca(ct, ang, ncomp=20, full_output=True, verbose=False, check_memory=False)
der = [o for o in out if torch.is_tensor(o) and o.ndim == 3 and o.shape[0] == 400][-1]
g = torch.randn_like(der)
ref = B.collapse(der, "median").clone()
for ch in (1, 0, 2, 4, 8, 16, 32, 64, 128):
    ctx.set_option("median_xcd_chunk", ch)
    same = torch.equal(torch.nan_to_num(B.collapse(der, "median"), nan=3.0), torch.nan_to_num(ref, nan=3.0))
    print("chunk %3d: real %.3f ms  gauss %.3f ms same %s" % (ch, t(lambda: B.collapse(der, "median")), t(lambda: B.collapse(g, "median")), same))
ctx.set_option("median_xcd_chunk", -1)
for tp in (0, 16, 32):
    ctx.set_option("median_tp", tp)
    print("default chunk, median_tp %2d: real %.3f ms  gauss %.3f ms" % (tp, t(lambda: B.collapse(der, "median")), t(lambda: B.collapse(g, "median"))))
