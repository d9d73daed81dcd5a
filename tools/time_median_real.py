"""Median collapse on the derotated residual cube of a real C2 call against the same call on Gaussian noise (with the same NaN mask)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi_device
def t(fn, reps=7):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
ct, ang = synth_adi_device(400, 512, seed=0)
out = pca(ct, ang, ncomp=20, full_output=True, verbose=False, check_memory=False)
der = [o for o in out if torch.is_tensor(o) and o.ndim == 3 and o.shape[0] == 400][-1]
nan = torch.isnan(der)
print("derotated residuals: NaN fraction %.3f, std %.3g, max|x| %.3g" % (float(nan.float().mean()), float(der[~nan].std()), float(der[~nan].abs().max())))
print("real data      : %.3f ms" % t(lambda: B.collapse(der, "median")))
g = torch.randn_like(der)
print("gaussian       : %.3f ms" % t(lambda: B.collapse(g, "median")))
g[nan] = float("nan")
print("gaussian + NaNs: %.3f ms" % t(lambda: B.collapse(g, "median")))
z = der.clone(); z[nan] = 0.0
print("real, NaN -> 0 : %.3f ms" % t(lambda: B.collapse(z, "median")))
