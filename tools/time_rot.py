"""Time cube_derotate alone (device-resident) for a few shapes / options:  python tools/time_rot.py N n [opt=val ...]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
N, n = int(sys.argv[1]), int(sys.argv[2])
ctx = B.get_context()
amax = 90.0
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    if k == "amax":                     # angles 0 .. amax (amax <= 44: no quarter turn, every gather of shear 1 is row-wise)
        amax = float(v)
    else:
        ctx.set_option(k, int(v))
cube = torch.randn(n, N, N, device="cuda")
ang = np.linspace(0, amax, n)
for _ in range(2): B.derotate(cube, ang)
torch.cuda.synchronize()
ctx.set_option("timing", 1); ctx.reset_timers()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): out = B.derotate(cube, ang)
e1.record(); torch.cuda.synchronize()
print(N, n, sys.argv[3:], "total %.3f ms" % (e0.elapsed_time(e1) / 5),
      {k: round(ctx.stage_ms(k) / 5, 3) for k in ("k_rot_s1", "k_rot_s2", "k_rot_s3", "k_rot_aux")})
