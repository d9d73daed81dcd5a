"""Do small kernels run at a reduced shader clock right after the power-capped legs?  10 s (default) of pipelined C2 calls, then
61 x 101 x 101 calls for 1.5 s while a thread samples the busy card's freq1_input / power1_input from sysfs every 2 ms; per-call
times and the clock are printed in 50 ms bins.   python tools/clock_after_cap.py [seconds_of_load]"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
cube, ang = synth_adi(400, 512, seed=1)
big = [torch.from_numpy(cube).cuda(), torch.from_numpy(cube).cuda()]
small_np, ang_s = synth_adi(61, 101, seed=11)
small = torch.from_numpy(small_np).cuda()
pin = torch.empty((101, 101), dtype=torch.float32).pin_memory()
for _ in range(30):
    pin.copy_(pca(small, ang_s, ncomp=5, verbose=False, check_memory=False), non_blocking=False)
pca(big[0], ang, ncomp=20, verbose=False, check_memory=False); torch.cuda.synchronize()
cards = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
def busiest():                      # (the box shows every GPU of the node in sysfs; ours is the one at the cap during the load)
    best, bd = -1, None
    for d in cards:
        try:
            p = int(open(os.path.join(d, "power1_input")).read())
        except Exception:
            continue
        if p > best:
            best, bd = p, d
    return bd, best
bd = None
samples, stop = [], False
def sampler():
    while not stop:
        if bd is not None:
            try:
                samples.append((time.perf_counter(), int(open(os.path.join(bd, "freq1_input")).read()) / 1e6, int(open(os.path.join(bd, "power1_input")).read()) / 1e6))
            except Exception:
                pass
        time.sleep(0.002)
th = threading.Thread(target=sampler); th.start()
streams = [torch.cuda.Stream() for _ in range(2)]
B.set_async(True)
t0 = time.perf_counter(); i = 0
while time.perf_counter() - t0 < secs:
    with torch.cuda.stream(streams[i % 2]):
        pca(big[i % 2], ang, ncomp=20, verbose=False, check_memory=False)
    i += 1
    if i % 8 == 0:
        streams[(i + 1) % 2].synchronize()
    if bd is None and time.perf_counter() - t0 > 0.4 * secs:
        bd, pw = busiest()
        print("card", bd, "power %.0f W under the load" % (pw / 1e6))
torch.cuda.synchronize()
B.set_async(False)
t_load_end = time.perf_counter()
calls = []
while time.perf_counter() - t_load_end < 1.5:
    t1 = time.perf_counter()
    pin.copy_(pca(small, ang_s, ncomp=5, verbose=False, check_memory=False), non_blocking=False)
    calls.append((t1 - t_load_end, time.perf_counter() - t1))
stop = True; th.join()
load = [s for s in samples if s[0] < t_load_end and s[0] > t_load_end - 2.0]
print("last 2 s of the load: clock %.0f MHz, power %.0f W (%d pipelined calls in %.1f s)" % (np.mean([s[1] for s in load]), np.mean([s[2] for s in load]), i, secs))
for b in range(30):
    lo, hi = b * 0.05, (b + 1) * 0.05
    cs = [c[1] for c in calls if lo <= c[0] < hi]
    ss = [s for s in samples if lo <= s[0] - t_load_end < hi]
    if cs and ss:
        print("t = %4.0f..%4.0f ms after the load: %3d calls, %.3f ms per call | clock %.0f MHz, power %.0f W" % (lo * 1e3, hi * 1e3, len(cs), np.mean(cs) * 1e3, np.mean([s[1] for s in ss]), np.mean([s[2] for s in ss])))
