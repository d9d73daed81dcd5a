"""Diagnostics of the medium-size differential case seed 108 (n 111, N 298, ncomp 12, temp-mean, collapse mean, mask 36)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import ref_cpu as O
from vip_amd.psfsub import pca
seed = 108
rng = np.random.default_rng(9000 + seed)
n = int(rng.integers(40, 260)); N = int(rng.integers(90, 300))
cube, _ = O.synth_adi(n, N, seed=int(rng.integers(1 << 30))); cube = cube.astype(np.float32)
ang = np.linspace(0, float(rng.uniform(40, 200)), n) if rng.integers(2) else np.sort(rng.uniform(-150, 150, n))
print(n, N, ang[:3], ang[-3:])
ref = np.load(os.path.join(ROOT, "tests/_fuzz_refs/ref_108.npz"))["ref"]
for kw in (dict(ncomp=12, scaling="temp-mean", collapse="mean", mask_center_px=36), dict(ncomp=12, scaling="temp-mean", collapse="median", mask_center_px=36),
           dict(ncomp=12, scaling="temp-mean", collapse="mean", mask_center_px=20), dict(ncomp=12, scaling="temp-mean", collapse="mean")):
    out = pca(cube, ang, verbose=False, **kw)
    if kw == dict(ncomp=12, scaling="temp-mean", collapse="mean", mask_center_px=36):
        d = np.abs(out - ref); d[~np.isfinite(d)] = 0
        yy, xx = np.mgrid[:N, :N]; cy = cx = N // 2
        r = np.hypot(yy - cy, xx - cx)
        bad = d > 1e-4
        print("case A: max %.3e at %s (radius %.1f); %d pixels over 1e-4, radii %.1f .. %.1f; ref there %.4f, out %.4f" % (
            d.max(), np.unravel_index(d.argmax(), d.shape), r.flat[d.argmax()], bad.sum(), r[bad].min() if bad.any() else -1, r[bad].max() if bad.any() else -1,
            ref.flat[d.argmax()], out.flat[d.argmax()]))
        fo = pca(cube, ang, verbose=False, full_output=True, **kw)
        ro = O.pca_fullframe(cube, ang, full_output=True, **kw)
        for nm, a, b in zip(("frame", "pcs", "recon", "res", "resder"), fo, ro):
            if nm == "pcs":
                continue
            dd = np.abs(np.asarray(a) - np.asarray(b)); dd[~np.isfinite(dd)] = 0
            print("   %-7s max|d| %.3e  (NaN pattern equal: %s)" % (nm, dd.max(), np.array_equal(np.isnan(a), np.isnan(b))))
            if nm == "resder":
                f = np.unravel_index(dd.argmax(), dd.shape)
                print("   resder worst at frame %d pixel (%d, %d) radius %.1f: ref %.5f out %.5f; angle %.3f" % (f[0], f[1], f[2], r[f[1], f[2]], b[f], a[f], ang[f[0]]))
                per_frame = dd.reshape(n, -1).max(1)
                print("   frames over 1e-4:", np.nonzero(per_frame > 1e-4)[0][:20], "of", n)
    else:
        r2 = O.pca_fullframe(cube, ang, **kw)
        dd = np.abs(out - r2); dd[~np.isfinite(dd)] = 0
        print(kw, "max|d| %.3e" % dd.max())
