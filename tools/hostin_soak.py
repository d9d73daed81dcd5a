"""Soak of the host-input entry: many pca(numpy) calls (plain and masked, two cube sizes alternating so that workspaces are
re-sized), every frame compared with the first one of its kind."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cases = []
for n, N, k, mpx in ((400, 512, 20, None), (300, 384, 6, 7), (257, 512, 9, None)):
    cube, ang = synth_adi(n, N, n)
    os.environ["VIPMI_HOSTIN"] = "0"
    ref = pca(cube, ang, ncomp=k, mask_center_px=mpx, verbose=False, check_memory=False)
    cases.append((cube, ang, k, mpx, ref))
os.environ["VIPMI_HOSTIN"] = "1"
t0 = time.perf_counter(); bad = 0
for i in range(steps):
    cube, ang, k, mpx, ref = cases[i % len(cases)]
    out = pca(cube, ang, ncomp=k, mask_center_px=mpx, verbose=False, check_memory=False)
    if not np.array_equal(out, ref, equal_nan=True):
        bad += 1
print("hostin soak: %d calls, %d mismatches, %.1f ms per call" % (steps, bad, (time.perf_counter() - t0) / steps * 1e3))
sys.exit(1 if bad else 0)
