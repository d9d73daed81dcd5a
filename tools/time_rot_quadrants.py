"""Derotation time against the quadrant of the angles (the rot90 of rotate_fft is folded into the gather of shear 1 and
into the placement of shear 3): 400 frames of 512 px with all angles in one quadrant."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from vip_amd import backend as B
N, n = 512, 400
ctx = B.get_context()
cube = torch.randn(n, N, N, device="cuda")
for name, lo, hi in (("q0 (|a| < 45)", -40, 40), ("q1 (45..135)", 50, 130), ("q2 (135..225)", 140, 220), ("q3 (225..315)", 230, 310)):
    ang = -np.linspace(lo, hi, n)          # cube_derotate rotates by -angle
    for _ in range(2): B.derotate(cube, ang)
    torch.cuda.synchronize()
    ctx.set_option("timing", 1); ctx.reset_timers()
    for _ in range(5): B.derotate(cube, ang)
    torch.cuda.synchronize()
    print(name, {k: round(ctx.stage_ms(k) / 5, 3) for k in ("k_rot_s1", "k_rot_s2", "k_rot_s3", "k_rot_aux")})
