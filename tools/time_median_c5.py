"""Median collapse of the derotated residuals of a C5-sized call (2000 x 1024^2, k = 50) against Gaussian noise of the same shape:
how much of the collapse time is the data's distribution (ties of the zero corners, heavy tails -> crowded bins, second levels).
python tools/time_median_c5.py [n N k]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi_device
n, N, k = ([int(a) for a in sys.argv[1:4]] + [2000, 1024, 50][len(sys.argv) - 1:])[:3]
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
ct, ang = synth_adi_device(n, N, seed=0)
out = pca(ct, ang, ncomp=k, full_output=True, verbose=False, check_memory=False)
der = [o for o in out if torch.is_tensor(o) and o.ndim == 3 and o.shape[0] == n][-1]
del out, ct
nan = torch.isnan(der)
zero = der == 0
print("derotated residuals %s: NaN fraction %.4f, exact zeros %.4f, std %.3g, max|x| %.3g" % (
    tuple(der.shape), float(nan.float().mean()), float(zero.float().mean()), float(der[~nan].std()), float(der[~nan].abs().max())))
gb = der.numel() * 4 / 1e9
for name, x in (("real data", der), ("gaussian", torch.randn_like(der)), ("gaussian^3 (heavy tails)", torch.randn_like(der) ** 3)):
    ms = t(lambda: B.collapse(x, "median"))
    print("%-26s: %.3f ms  (%.2f TB/s)" % (name, ms, gb / ms))
z = der.clone(); z[zero] = torch.randn_like(z)[zero] * 1e-3
print("%-26s: %.3f ms" % ("real, zeros -> tiny noise", t(lambda: B.collapse(z, "median"))))
