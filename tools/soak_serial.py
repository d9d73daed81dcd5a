"""Soak of the SYNCHRONOUS path (one-XCD eigensolver exchange, one staging slot): many un-pipelined calls from two threads at once
plus a single-thread loop; every frame bit-identical to the first.   python tools/soak_serial.py [calls]"""
import sys, os, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi_device
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
OPTS = {o.split('=')[0]: int(o.split('=')[1]) for o in sys.argv[2:]}
from vip_amd import backend as B
def setopts():
    c = B.get_context()
    for a, b in OPTS.items(): c.set_option(a, b)
ct, ang = synth_adi_device(400, 512, seed=0)
setopts()
ref = pca(ct, ang, ncomp=20, verbose=False, check_memory=False).clone()
t0 = time.perf_counter()
for i in range(steps):
    assert torch.equal(pca(ct, ang, ncomp=20, verbose=False, check_memory=False), ref), i
print("single thread: %d calls identical, %.2f ms per call" % (steps, (time.perf_counter() - t0) / steps * 1e3))
errs = []
def work(k):
    try:
        with torch.cuda.stream(torch.cuda.Stream()):
            setopts()
            for i in range(steps // 2):
                o = pca(ct, ang, ncomp=20, verbose=False, check_memory=False)
                torch.cuda.current_stream().synchronize()
                if not torch.equal(o, ref): errs.append((k, i, float((o - ref).abs().max())))
    except Exception as e:      # noqa: BLE001
        errs.append(e)
th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]
assert not errs, errs[:5]
print("three threads: %d calls each identical, %.2f ms per call overall" % (steps // 2, (time.perf_counter() - t0) / (3 * (steps // 2)) * 1e3))
