"""Median collapse alone for a few (n, N): the tile kernel (median_reg=0, tile widths median_tp) against the register kernel
(median_reg=1, default from 129 to 2048 frames).   python tools/time_median.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vip_amd import backend as B
ctx = B.get_context()
def t(fn, reps=7):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
for n, N in ((400, 512), (2000, 1024), (1000, 512), (200, 512), (200, 256), (130, 512), (600, 512)):
    cube = torch.randn(n, N, N, device="cuda")
    cube[torch.rand_like(cube) < 0.01] = float("nan")
    ref = None
    for reg, tp in ((0, 0), (0, 32), (1, 0)):
        ctx.set_option("median_reg", reg); ctx.set_option("median_tp", tp)
        try:
            out = B.collapse(cube, "median")
            ms = t(lambda: B.collapse(cube, "median"))
        except Exception as e:
            print("n %d N %d reg %d tp %d: %s" % (n, N, reg, tp, str(e)[:60])); continue
        if ref is None: ref = out.clone()
        same = bool(torch.equal(torch.nan_to_num(out, nan=7.5), torch.nan_to_num(ref, nan=7.5)))
        print("n %4d N %4d median_reg %d median_tp %2d: %.3f ms  (%.2f TB/s)  same %s" % (n, N, reg, tp, ms, cube.numel() * 4 / ms / 1e9, same))
    ctx.set_option("median_reg", 1); ctx.set_option("median_tp", 0)
    tm = t(lambda: B.collapse(cube, "trimmean", trim_n=n // 2))
    print("n %4d N %4d trimmean (register kernel): %.3f ms" % (n, N, tm))
    del cube
