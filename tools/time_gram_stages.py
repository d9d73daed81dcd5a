"""Gram stage alone at C2 and C5 size (stage timer), best of 5."""
import sys, os
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
for n, P in ((400, 512 * 512), (2000, 1024 * 1024)):
    x = torch.randn(n, P, device="cuda"); G = torch.empty(n, n, dtype=torch.float64, device="cuda")
    print("n %d P %d: gram %.3f ms" % (n, P, t(lambda: ctx.call("vipmi_gram_f32", B.ptr(x), n, P, P, B.ptr(G)))))
    del x
