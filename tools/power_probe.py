"""Socket power and shader clock while one stage runs in a loop for a few seconds (sysfs hwmon; amd-smi as the fall-back).
usage: python tools/power_probe.py rot512 | rot256 | gram | idle [seconds]     (on the GPU box)"""
import glob, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B

what = sys.argv[1] if len(sys.argv) > 1 else "rot512"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0


def hwmon():
    """the card that draws the most power (the box shows every GPU of the node in sysfs; ours is the busy one)"""
    best = {}
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        cur = {"card": d.split("/")[4]}
        for name in ("power1_average", "power1_input", "freq1_input", "power1_cap"):
            p = os.path.join(d, name)
            if os.path.exists(p):
                try:
                    cur[name] = int(open(p).read())
                except Exception:
                    pass
        if cur.get("power1_input", cur.get("power1_average", 0)) > best.get("power1_input", best.get("power1_average", -1)):
            best = cur
    out = best
    for p in glob.glob("/sys/class/drm/%s/device/pp_dpm_sclk" % out.get("card", "card*")):
        try:
            cur = [l for l in open(p).read().splitlines() if l.strip().endswith("*")]
            if cur:
                out["sclk"] = cur[0]
        except Exception:
            pass
    return out


samples, stop = [], False


def sampler():
    while not stop:
        samples.append((time.perf_counter(), hwmon()))
        time.sleep(0.02)


if what.startswith("rot"):
    N = int(what[3:]); n = {512: 400, 256: 1600, 1024: 100}[N]
    cube = torch.randn(n, N, N, device="cuda"); ang = np.linspace(0, 90, n)
    fn = lambda: B.derotate(cube, ang)
elif what == "gram":
    M = torch.randn(400, 512 * 512, device="cuda")
    fn = lambda: B.gram(M)
else:
    fn = lambda: time.sleep(0.01)
for _ in range(3): fn()
torch.cuda.synchronize()
print("idle", hwmon())
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); it = 0
while time.perf_counter() - t0 < secs:
    for _ in range(10): fn()
    torch.cuda.synchronize(); it += 10
t1 = time.perf_counter()
stop = True; th.join()
mid = [s for t, s in samples if t0 + 0.5 < t < t1 - 0.1]
print(what, "%.3f ms per call over %.1f s" % ((t1 - t0) / it * 1e3, t1 - t0), "samples", len(mid))
for key in ("power1_average", "power1_input", "freq1_input"):
    v = [s[key] for s in mid if key in s]
    if v:
        print("  %-16s mean %.1f  min %.1f  max %.1f   (W / MHz)" % (key, np.mean(v) / 1e6, np.min(v) / 1e6, np.max(v) / 1e6))
print("  cards", sorted(set(s.get("card") for s in mid)), "cap", mid[-1].get("power1_cap") if mid else None, "sclk", sorted(set(s.get("sclk", "?") for s in mid))[:6])
if not mid or not any("power1_average" in s or "power1_input" in s for s in mid):
    try:
        print(subprocess.run(["amd-smi", "metric", "-p", "-c"], capture_output=True, text=True, timeout=20).stdout[-1500:])
    except Exception as e:
        print("amd-smi:", e)
