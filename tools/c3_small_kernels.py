"""C3 (annular PCA) a few times for a kernel trace of its small kernels (gather / scatter / gram_reduce / coeff_range) + wall time."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca_annular
cube, ang = synth_adi(400, 512, 0); ct = torch.from_numpy(cube).cuda()
f = lambda: pca_annular(ct, ang, asize=32, ncomp=10, fwhm=4, delta_rot=(0.1, 1), verbose=False)
ref = f().cpu().numpy(); torch.cuda.synchronize()
ts = []
for rep in range(24):
    t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print("C3 per call (ms):", " ".join("%.1f" % x for x in ts), "| median %.2f" % float(np.median(ts)), flush=True)
import gc
gc.disable(); ts = []
for rep in range(40):
    t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
gc.enable()
print("gc disabled     :", " ".join("%.1f" % x for x in ts), "| median %.2f max %.1f" % (float(np.median(ts)), max(ts)), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "c3_frame.npy"), ref)
