"""How the host waits for a small result (61 x 101 x 101 resident cube, frame into pinned memory; and the C2 call): torch's
stream.synchronize() (hipStreamSynchronize: may sleep on an interrupt), a spin on event.query(), a blocking .cpu().
   python tools/sync_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi
import gc

def run(n, N, k, reps, warm):
    cube, ang = synth_adi(n, N, seed=11)
    ct = torch.from_numpy(cube).cuda()
    pin = torch.empty((N, N), dtype=torch.float32).pin_memory()
    ev = torch.cuda.Event()
    def a():
        fr = pca(ct, ang, ncomp=k, verbose=False, check_memory=False)
        pin.copy_(fr, non_blocking=True)
        torch.cuda.current_stream().synchronize()
    def b():
        fr = pca(ct, ang, ncomp=k, verbose=False, check_memory=False)
        pin.copy_(fr, non_blocking=True)
        ev.record()
        while not ev.query():
            pass
    def c():
        pca(ct, ang, ncomp=k, verbose=False, check_memory=False).cpu()
    def d():
        fr = pca(ct, ang, ncomp=k, verbose=False, check_memory=False)
        pin.copy_(fr, non_blocking=False)
    gc.collect(); gc.freeze()
    for name, fn in (("stream.synchronize", a), ("event.query spin", b), (".cpu()", c), ("blocking copy_ to pinned", d),
                     ("stream.synchronize", a), ("event.query spin", b)):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e3
        print("%dx%dx%d %-26s mean %.3f ms  median %.3f  min %.3f  p90 %.3f" % (n, N, N, name, ts.mean(), np.median(ts), ts.min(), np.percentile(ts, 90)), flush=True)

run(61, 101, 5, 200, 60)
run(400, 512, 20, 30, 5)
