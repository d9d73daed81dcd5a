"""Median collapse against np.nanmedian (bit for bit) on awkward cubes, then the timings that matter: the derotated residuals of a C2
call, Gaussian noise at C2 and C5 size.   python tools/median_check.py [quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, warnings
from vip_amd import backend as B
warnings.filterwarnings("ignore")
def t(fn, reps=7):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
rng = np.random.default_rng(5)
bad = 0
for n in (1, 2, 3, 63, 64, 65, 130, 257, 400, 448, 449, 600, 1030, 1500, 2048, 2100):
    P = (24, 40)
    cases = {}
    x = rng.standard_normal((n,) + P).astype(np.float32); cases["normal"] = x
    y = x.copy(); y[rng.random(y.shape) < 0.3] = np.nan; y[:, 0, :5] = np.nan; cases["nan30"] = y
    cases["ties"] = np.round(x * 2).astype(np.float32)
    z = np.full_like(x, 3.25); z[:, 1] = -0.0; z[::2, 2] = 0.0; z[1::2, 2] = -0.0; cases["const"] = z
    o = x.copy(); o[0] = 1e30; o[-1] = -1e30; o[n // 2, :, ::3] = np.inf; cases["outliers"] = o
    c = (7000 + 45 * x).astype(np.float32); cases["counts"] = c
    d = x.copy() * 1e-41; cases["denormal"] = d.astype(np.float32)
    w = np.where(rng.random(x.shape) < 0.9, np.float32(1.5), x).astype(np.float32); cases["crowded"] = w
    for name, a in cases.items():
        ref = np.nanmedian(a, axis=0)
        got = B.collapse(torch.from_numpy(a).cuda(), "median").cpu().numpy()
        same = np.array_equal(ref, got, equal_nan=True)
        if not same:
            bad += 1
            i = np.argwhere(~((ref == got) | (np.isnan(ref) & np.isnan(got))))[0]
            print("MISMATCH n %d %s at %s: ref %r got %r (%d pixels)" % (n, name, i, ref[tuple(i)], got[tuple(i)], int((~((ref == got) | (np.isnan(ref) & np.isnan(got)))).sum())))
print("bit-exact cases failed: %d" % bad)
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(1 if bad else 0)
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi_device
ct, ang = synth_adi_device(400, 512, seed=0)
out = pca(ct, ang, ncomp=20, full_output=True, verbose=False, check_memory=False)
der = [o for o in out if torch.is_tensor(o) and o.ndim == 3 and o.shape[0] == 400][-1]
ref = np.nanmedian(der.cpu().numpy(), axis=0)
got = B.collapse(der, "median").cpu().numpy()
print("C2 residuals bit-exact vs np.nanmedian:", np.array_equal(ref, got, equal_nan=True))
print("C2 real residuals : %.3f ms" % t(lambda: B.collapse(der, "median")))
g = torch.randn_like(der)
print("C2 gaussian       : %.3f ms" % t(lambda: B.collapse(g, "median")))
print("C2 trimmean       : %.3f ms" % t(lambda: B.collapse(der, "trimmean", trim_n=200)))
del ct, out, der, g
g = torch.randn(2000, 1024, 1024, device="cuda")
print("C5 gaussian       : %.3f ms" % t(lambda: B.collapse(g, "median"), 4))
# residual-like: heavy-tailed (a few bright frames per pixel), NaN corners
g *= (1 + 5 * (torch.rand(2000, 1, 1, device="cuda") < 0.02))
yy, xx = torch.meshgrid(torch.arange(1024, device="cuda"), torch.arange(1024, device="cuda"), indexing="ij")
g[:200, ((yy - 512) ** 2 + (xx - 512) ** 2) > 500 ** 2] = float("nan")
print("C5 heavy tails+NaN: %.3f ms" % t(lambda: B.collapse(g, "median"), 4))
r = np.nanmedian(g[:, 500:520, :64].cpu().numpy(), axis=0); q = B.collapse(g, "median")[500:520, :64].cpu().numpy()
print("C5 sample bit-exact:", np.array_equal(r, q, equal_nan=True))
sys.exit(1 if bad else 0)
