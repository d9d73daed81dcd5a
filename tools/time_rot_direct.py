"""Time the generic (non power-of-two) derotation path: python tools/time_rot_direct.py  (VIPMI_OPTS=rot_conv=0: the
direct correlations instead of the convolution passes)"""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from vip_amd import backend as B
ctx = B.get_context()
for N in (101, 129, 201, 255, 301, 401, 511, 512):
    n = 100
    cube = torch.randn(n, N, N, device="cuda"); ang = np.linspace(0, 350, n)
    for _ in range(2): B.derotate(cube, ang, method="direct" if N == 512 else "auto")
    torch.cuda.synchronize()
    ctx.set_option("timing", 1); ctx.reset_timers()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): B.derotate(cube, ang, method="direct" if N == 512 else "auto")
    e1.record(); torch.cuda.synchronize()
    print("N=%d: %.3f ms per %d frames   " % (N, e0.elapsed_time(e1) / 3, n),
          {s: round(ctx.stage_ms(s) / 3, 3) for s in ("k_rot_s1", "k_rot_aux", "k_rot_s2", "k_rot_s3") if ctx.stage_count(s)})
    ctx.set_option("timing", 0)
