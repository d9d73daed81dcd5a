"""Gram matrix on the int8 matrix cores (csrc/gram_i8.hip) against the float64-MFMA kernel and a float64 torch product:
accuracy, time, and what it does to the leading subspace / the final frame of pca().   python tools/gram_i8_check.py [n N k]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vip_amd import backend as B
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi, synth_adi_device

n, N, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (400, 512, 20)
if n * N * N > 2 ** 28:
    ct, ang = synth_adi_device(n, N, seed=0)
else:
    cube, ang = synth_adi(n, N, seed=0)
    ct = torch.from_numpy(cube).cuda()
M = ct.reshape(n, -1)
ctx = B.get_context()
chunk = 1 << 16
Gex = torch.zeros((n, n), dtype=torch.float64, device="cuda")
for c0 in range(0, M.shape[1], chunk):
    Mc = M[:, c0:c0 + chunk].double()
    Gex += Mc @ Mc.T
gmax = float(Gex.abs().max())


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


w, V = torch.linalg.eigh(Gex)
Vk = V[:, -k:]
res = {}
for mode in (0, 1, 2):
    ctx.set_option("gram_i8", mode)
    G = B.gram(M)
    err = float((G - Gex).abs().max()) / gmax
    asym = float((G - G.T).abs().max()) / gmax
    ms = timed(lambda: B.gram(M))
    w2, V2 = torch.linalg.eigh(G)
    sin = float(torch.linalg.svdvals(V[:, :-k].T @ V2[:, -k:]).max())
    fr = pca(ct, ang, ncomp=k, verbose=False, check_memory=False)
    res[mode] = fr
    print("gram_i8=%d: max|dG|/max|G| %.2e  asym %.1e  %.3f ms   sin(theta_k) vs float64 %.2e   frame max|d| vs mode 0: %.2e" % (
        mode, err, asym, ms, sin, float((fr - res[0]).abs()[torch.isfinite(fr)].max())))
ctx.set_option("gram_i8", 0)
