"""61x101x101 (the tutorial's shape): why does a resident call take longer than a numpy call?  (round 6 probe)"""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca
cube, ang = synth_adi(61, 101, seed=11); ct = torch.from_numpy(cube).cuda()
def lat(fn, reps=50):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for rnd in range(3):
    print("resident + .cpu(): %.3f ms   resident, no copy: %.3f ms   numpy in/out: %.3f ms" % (
        lat(lambda: pca(ct, ang, ncomp=5, verbose=False, check_memory=False).cpu()),
        lat(lambda: pca(ct, ang, ncomp=5, verbose=False, check_memory=False)),
        lat(lambda: pca(cube, ang, ncomp=5, verbose=False, check_memory=False))))
ctx = B.get_context(); ctx.set_option("timing", 3); ctx.reset_timers()
for _ in range(20): pca(ct, ang, ncomp=5, verbose=False, check_memory=False).cpu()
print("host ms per stage (timing=3):", {s: round(ctx.stage_ms(s) / 20, 3) for s in ("gram", "eigh", "project", "derotate", "collapse")})
ctx.set_option("timing", 0)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50): pca(ct, ang, ncomp=5, verbose=False, check_memory=False).cpu()
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
