"""matrix_scaling modes and the mask at C2 size, kernel time (best of 7)."""
import sys, os
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
x = torch.randn(400, 512 * 512, device="cuda"); out = torch.empty_like(x)
mask = (torch.rand(512 * 512, device="cuda") < 0.1).to(torch.uint8)
for mode, name in ((1, "temp-mean"), (2, "temp-standard"), (3, "spat-mean"), (4, "spat-standard")):
    print("%-14s %.3f ms" % (name, t(lambda: ctx.call("vipmi_scale_f32", B.ptr(x), B.ptr(out), 400, 512 * 512, mode))))
print("%-14s %.3f ms" % ("mask", t(lambda: ctx.call("vipmi_apply_mask_f32", B.ptr(x), B.ptr(out), 400, 512 * 512, B.ptr(mask), 0.0))))
