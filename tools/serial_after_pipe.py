"""bisect: serial-call latency after a pipelined phase (bench.py's latency leg), per call, with / without the recovery launch"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi_device
ct, ang = synth_adi_device(400, 512, seed=0)
c2, _ = synth_adi_device(400, 512, seed=1)
def serial(tag, reps=8):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); pca(ct, ang, ncomp=20, verbose=False, check_memory=False).cpu(); ts.append((time.perf_counter() - t0) * 1e3)
    print(tag, " ".join("%.2f" % t for t in ts), flush=True)
serial("fresh serial     ")
if "pipe" in sys.argv:
    streams = [torch.cuda.Stream() for _ in range(2)]
    B.set_async(True)
    for i in range(20):
        with torch.cuda.stream(streams[i % 2]):
            pca((ct, c2)[i % 2], ang, ncomp=20, verbose=False, check_memory=False)
    torch.cuda.synchronize(); B.check_deferred(); B.set_async(False)
    serial("after pipelined  ")
ctx = B.get_context()
ctx.set_option("eigh_recover", 0); serial("eigh_recover=0   ")
ctx.set_option("eigh_recover", 1); serial("eigh_recover=1   ")
