"""How crowded is the bin that holds the median under the kernel's first-level binning (256 linear bins between a pixel's
minimum and maximum)?  Derotated residuals of a real call against Gaussian noise.  python tools/median_bins_probe.py [n N k]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi_device
n, N, k = ([int(a) for a in sys.argv[1:4]] + [2000, 1024, 50][len(sys.argv) - 1:])[:3]
ct, ang = synth_adi_device(n, N, seed=0)
out = pca(ct, ang, ncomp=k, full_output=True, verbose=False, check_memory=False)
der = [o for o in out if torch.is_tensor(o) and o.ndim == 3 and o.shape[0] == n][-1]
del out, ct
def probe(name, x):
    x = x.reshape(n, -1)[:, ::7].contiguous()            # every 7th pixel
    lo, hi = x.min(0).values, x.max(0).values
    med = x.median(0).values
    w = (hi - lo) / 256
    b = torch.clamp(((med - lo) / w).floor(), 0, 255)
    cnt = ((x >= lo + b * w) & (x < lo + (b + 1) * w)).sum(0).float()
    sd = x.std(0)
    kurt = (((x - x.mean(0)) / sd) ** 4).mean(0)
    print("%-10s keys in the median's bin: mean %.1f, p50 %.0f, p90 %.0f, p99 %.0f, max %.0f; share of pixels with more than 64: %.3f; "
          "range / std: median %.1f, p99 %.1f; kurtosis median %.2f p99 %.1f" % (
              name, cnt.mean(), cnt.quantile(0.5), cnt.quantile(0.9), cnt.quantile(0.99), cnt.max(), (cnt > 64).float().mean(),
              ((hi - lo) / sd).median(), ((hi - lo) / sd).quantile(0.99), kurt.median(), kurt.quantile(0.99)))
probe("real", der)
probe("gaussian", torch.randn_like(der))
