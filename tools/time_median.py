"""Median collapse alone for a few (n, N) and tile widths (option median_tp): python tools/time_median.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vip_amd import backend as B
ctx = B.get_context()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
for n, N in ((400, 512), (2000, 1024), (1000, 512), (200, 512)):
    cube = torch.randn(n, N, N, device="cuda")
    ref = None
    for tp in (0, 8, 16, 32):
        ctx.set_option("median_tp", tp)
        try:
            out = B.collapse(cube, "median")
            ms = t(lambda: B.collapse(cube, "median"))
        except Exception as e:
            print("n %d N %d tp %d: %s" % (n, N, tp, str(e)[:60])); continue
        if ref is None: ref = out.clone()
        print("n %4d N %4d median_tp %2d: %.3f ms  (%.2f TB/s)  same %s" % (n, N, tp, ms, cube.numel() * 4 / ms / 1e9, bool(torch.equal(out, ref))))
    del cube
ctx.set_option("median_tp", 0)
