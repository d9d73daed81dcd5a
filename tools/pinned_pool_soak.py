"""full_output results in pinned blocks: a loop that drops its results reuses the same blocks (resident set and the pool's own
count stay flat), a loop that keeps them grows and falls back to pageable memory beyond VIPMI_PINNED_OUT_MB.  (round 6)"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, psutil, torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca, pca_annular
proc = psutil.Process()
cube, ang = synth_adi(200, 256, seed=1)            # 52 MB per cube-sized result
rss = lambda: proc.memory_info().rss / 2 ** 20
out = pca(cube, ang, ncomp=8, full_output=True, verbose=False); del out; gc.collect()
r0 = rss(); t0 = time.perf_counter()
for i in range(150):
    out = pca(cube, ang, ncomp=8, full_output=True, verbose=False)
    if i % 3 == 0:
        co, cd, fr = pca_annular(cube, ang, ncomp=3, asize=32, fwhm=4, full_output=True, verbose=False)
dt = (time.perf_counter() - t0) / 150
del out, co, cd, fr; gc.collect()
print("150 iterations dropping their results: %.2f ms per iteration, resident set %.0f -> %.0f MB, pinned bytes alive now %d" % (dt * 1e3, r0, rss(), B._pin_out["bytes"]))
os.environ["VIPMI_PINNED_OUT_MB"] = "512"
keep = [pca(cube, ang, ncomp=8, full_output=True, verbose=False) for _ in range(6)]
print("6 kept results under a 512 MB cap: pinned bytes alive %.0f MB (<= 512), every array intact: %s" % (
    B._pin_out["bytes"] / 2 ** 20, all(np.array_equal(k[3], keep[0][3]) for k in keep)))
