"""Where the host time of one un-pipelined pca() call goes (cube resident in HBM): wall time per call, sum of the stage
timers, and the Python profile of the front.   python tools/host_overhead.py [n N k]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vip_amd import backend as B
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi

n, N, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (400, 512, 20)
cube, ang = synth_adi(n, N, seed=0)
ct = torch.from_numpy(cube).cuda()
ctx = B.get_context()


def call():
    return pca(ct, ang, ncomp=k, verbose=False, check_memory=False).cpu()


for _ in range(3):
    call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    call()
wall = (time.perf_counter() - t0) / 20 * 1e3
ctx.set_option("timing", 1)
ctx.reset_timers()
for _ in range(5):
    call()
stages = {s: ctx.stage_ms(s) / 5 for s in ("scale", "gram", "eigh", "project", "derotate", "collapse") if ctx.stage_count(s)}
ctx.set_option("timing", 0)
print("wall per call %.3f ms; stage timers %s sum %.3f ms -> host / gaps %.3f ms" % (wall, {a: round(b, 3) for a, b in stages.items()},
                                                                                    sum(stages.values()), wall - sum(stages.values())))
# host time of the front alone: enqueue without waiting (asynchronous mode never synchronises)
B.set_async(True)
torch.cuda.synchronize()
t0 = time.perf_counter()
outs = [pca(ct, ang, ncomp=k, verbose=False, check_memory=False) for _ in range(20)]
enq = (time.perf_counter() - t0) / 20 * 1e3
torch.cuda.synchronize()
B.check_deferred()
B.set_async(False)
print("host time to enqueue one call (asynchronous mode): %.3f ms" % enq)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    call()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
