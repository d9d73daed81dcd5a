"""Does the speed of a 20-step pipelined pass depend on the TOTAL number of calls before it or on the length of the last burst?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi_device
pattern = sys.argv[1].split(",")          # bursts before the measured pass, e.g. 7  or 7,7,7 or 20; mNNN = NNN ms of torch matmuls
# on one stream (GPU load without our library), qNNN = NNN tiny kernels queued on each of the two streams (queue depth without load)
n, N, k, depth, K = 400, 512, 20, 2, 20
cubes = [synth_adi_device(n, N, seed=s)[0] for s in range(depth)]
ang = np.linspace(0, 90, n)
streams = [torch.cuda.Stream() for _ in range(depth)]
pinned = [torch.empty((N, N), dtype=torch.float32).pin_memory() for _ in range(K)]
B.set_async(True)
for o in sys.argv[2:]:
    a_, b_ = o.split('=')
    os.environ['VIPMI_OPT_' + a_] = b_
OPTS = {o.split('=')[0]: int(o.split('=')[1]) for o in sys.argv[2:]}
def run(m, rec=None):
    for i in range(m):
        with torch.cuda.stream(streams[i % depth]):
            if OPTS:
                c_ = B.get_context()
                for a_, b_ in OPTS.items(): c_.set_option(a_, b_)
            fr = pca(cubes[i % depth], ang, ncomp=k, verbose=False, check_memory=False)
            pinned[i % K].copy_(fr, non_blocking=True)
            if rec is not None:
                e = torch.cuda.Event(enable_timing=True); e.record(); rec.append(e)
A_ = torch.randn(4096, 4096, device="cuda")
for b in pattern:
    if b[0] == "m":
        t_ = time.perf_counter()
        while time.perf_counter() - t_ < float(b[1:]) * 1e-3:
            for _ in range(8): A_ @ A_
            if torch.cuda.current_stream().query(): pass
        torch.cuda.synchronize()
    elif b[0] == "t":                                    # tNN: NN pipelined pca() calls on TINY cubes (50 x 128 x 128)
        if "tiny" not in globals():
            tiny = [synth_adi_device(50, 128, seed=9 + s_)[0] for s_ in range(depth)]; tang = np.linspace(0, 90, 50)
        for i in range(int(b[1:])):
            with torch.cuda.stream(streams[i % depth]):
                fr = pca(tiny[i % depth], tang, ncomp=5, verbose=False, check_memory=False)
                pinned[i % K][:128, :128].copy_(fr, non_blocking=True)
        torch.cuda.synchronize()
    elif b[0] == "q":
        z_ = torch.zeros(64, device="cuda")
        for st in streams:
            with torch.cuda.stream(st):
                for _ in range(int(b[1:])): z_.add_(1.0)
        torch.cuda.synchronize()
    else:
        run(int(b)); torch.cuda.synchronize()
rec = []
e0 = torch.cuda.Event(enable_timing=True); e0.record()
run(K, rec); torch.cuda.synchronize()
ends = [e0.elapsed_time(e) for e in rec]
print("bursts %s -> %.1f ms; periods: %s" % (pattern, ends[-1], " ".join("%.1f" % (b - a) for a, b in zip(ends, ends[1:]))))
B.check_deferred(); B.set_async(False)
