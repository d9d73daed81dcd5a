"""Time the generic (non power-of-two) derotation path: python tools/time_rot_direct.py"""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from vip_amd import backend as B
for N in (101, 201, 301, 511):
    n = 100
    cube = torch.randn(n, N, N, device="cuda"); ang = np.linspace(0, 350, n)
    for _ in range(2): B.derotate(cube, ang)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): B.derotate(cube, ang)
    e1.record(); torch.cuda.synchronize()
    print("N=%d: %.3f ms per %d frames" % (N, e0.elapsed_time(e1) / 3, n))
