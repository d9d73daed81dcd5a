"""cube_derotate(numpy cube): upload / rotate / download pipelined in blocks of frames (VIPMI_HOSTIN) against one after the other."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd.preproc import cube_derotate
rng = np.random.default_rng(0)
gc.collect(); gc.freeze()
for n, N, dt in ((400, 512, np.float32), (400, 512, np.float64), (100, 511, np.float32), (37, 1024, np.float32)):
    cube = rng.standard_normal((n, N, N)).astype(dt); cube[:, :3, :5] = np.nan
    ang = np.linspace(-170, 175, n); res = {}
    for h in ("0", "1", "0", "1"):
        os.environ["VIPMI_HOSTIN"] = h
        cube_derotate(cube, ang); ts = []
        for _ in range(4):
            t0 = time.perf_counter(); out = cube_derotate(cube, ang); ts.append((time.perf_counter() - t0) * 1e3)
        res[h] = out
        print("%d x %d^2 %s hostin %s: %.1f ms (min of 4)" % (n, N, np.dtype(dt).name, h, min(ts)), flush=True)
    print("   identical: %s, dtype %s" % (np.array_equal(res["0"], res["1"], equal_nan=True), res["1"].dtype))
