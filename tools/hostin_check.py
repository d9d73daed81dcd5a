"""pca(numpy float32 cube) through the host-input fused entry (Gram under the upload) against upload-then-call: frames and
full_output arrays bit-identical, time per call.   python tools/hostin_check.py"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca
for n, N, k, mpx in ((400, 512, 20, None), (300, 512, 7, 8), (257, 384, 5, None), (520, 320, 30, 5)):
    cube, ang = synth_adi(n, N, 1)
    outs = {}
    for h in ("0", "1"):
        os.environ["VIPMI_HOSTIN"] = h
        outs[h] = pca(cube, ang, ncomp=k, mask_center_px=mpx, verbose=False, check_memory=False)
    same = np.array_equal(outs["0"], outs["1"], equal_nan=True)
    full = {}
    for h in ("0", "1"):
        os.environ["VIPMI_HOSTIN"] = h
        full[h] = pca(cube, ang, ncomp=k, mask_center_px=mpx, verbose=False, check_memory=False, full_output=True)
    same_full = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(full["0"], full["1"]))
    print("n %d N %d k %d mask %s: frame identical %s, full_output identical %s" % (n, N, k, mpx, same, same_full), flush=True)
cube, ang = synth_adi(400, 512, 0)
gc.collect(); gc.freeze()
for h in ("0", "1", "0", "1"):
    os.environ["VIPMI_HOSTIN"] = h
    pca(cube, ang, ncomp=20, verbose=False, check_memory=False); ts = []
    for _ in range(12):
        t0 = time.perf_counter(); pca(cube, ang, ncomp=20, verbose=False, check_memory=False); ts.append((time.perf_counter() - t0) * 1e3)
    print("pca(numpy C2 cube), hostin %s: median %.2f ms, min %.2f" % (h, float(np.median(ts)), min(ts)), flush=True)
