"""Where a 61 x 101 x 101 pca() call spends its host time (the bench's tutorial-shape leg is 0.45 ms on most boxes and 1.2 ms on
some, with the same 0.41 ms of device time): variants of handing the frame over, enqueue-only time, per-stage host clocks.
   python tools/small_call_breakdown.py"""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi

n, N, k = 61, 101, 5
cube, ang = synth_adi(n, N, seed=11)
ct = torch.from_numpy(cube).cuda()
pin = torch.empty((N, N), dtype=torch.float32).pin_memory()
ctx = B.get_context()
print("cpus %d, affinity %d, loadavg %s" % (os.cpu_count(), len(os.sched_getaffinity(0)), open("/proc/loadavg").read().strip()))
try:
    mhz = [float(l.split(":")[1]) for l in open("/proc/cpuinfo") if l.startswith("cpu MHz")]
    print("cpu MHz min %.0f max %.0f; model: %s" % (min(mhz), max(mhz), [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]))
except Exception as e:
    print("cpuinfo:", e)
gc.collect(); gc.freeze()

def timed(name, fn, reps=200, warm=60, sync_each=True):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    t00 = time.perf_counter()
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t00) / reps * 1e3
    ts = np.array(ts) * 1e3
    print("%-44s per call %.3f ms (host side of the call: median %.3f, p90 %.3f, max %.3f)" % (name, tot, np.median(ts), np.percentile(ts, 90), ts.max()), flush=True)

def A():
    pin.copy_(pca(ct, ang, ncomp=k, verbose=False, check_memory=False), non_blocking=False)
def Bv():
    pca(ct, ang, ncomp=k, verbose=False, check_memory=False); torch.cuda.synchronize()
def C():
    pca(ct, ang, ncomp=k, verbose=False, check_memory=False).cpu()
def D():
    pca(cube, ang, ncomp=k, verbose=False, check_memory=False)
def E():
    pca(ct, ang, ncomp=k, verbose=False, check_memory=False)
for rnd in range(2):
    timed("A resident, blocking copy to pinned", A)
    timed("B resident, device synchronize, no copy", Bv)
    timed("C resident, .cpu()", C)
    timed("D numpy in, numpy out", D)
    timed("E resident, enqueue only (no wait per call)", E)
ctx.set_option("timing", 3); ctx.reset_timers()
for _ in range(50):
    A()
torch.cuda.synchronize()
print("host ms per stage (timing = 3):", {s: round(ctx.stage_ms(s) / 50, 4) for s in ("scale", "gram", "eigh", "project", "derotate", "collapse") if ctx.stage_count(s) > 0})
ctx.set_option("timing", 1); ctx.reset_timers()
for _ in range(50):
    A()
torch.cuda.synchronize()
print("device ms per stage (timing = 1):", {s: round(ctx.stage_ms(s) / 50, 4) for s in ("scale", "gram", "eigh", "project", "derotate", "collapse") if ctx.stage_count(s) > 0})
ctx.set_option("timing", 0)
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    A()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(8); print(s.getvalue()[:1800])
