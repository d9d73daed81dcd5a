"""int8 Gram against the float64-MFMA Gram over problem sizes (single and batched): time and accuracy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vip_amd import backend as B
ctx = B.get_context()
ctx.set_option("gram_i8_min_n", 16)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
g = torch.Generator(device="cuda").manual_seed(1)
for batch, n, P in ((1, 32, 16384), (1, 50, 16384), (1, 64, 65536), (1, 100, 10201), (1, 128, 65536), (1, 200, 65536), (1, 256, 262144), (1, 400, 262144),
                    (1, 640, 65536), (1, 1000, 65536), (39, 200, 65536), (8, 196, 30000), (200, 39, 65536), (16, 100, 16384)):
    M = torch.randn((batch, n, P), device="cuda", generator=g) * torch.linspace(0.1, 10, P, device="cuda")
    ref = torch.bmm(M.double(), M.double().transpose(1, 2))
    out = {}
    for mode in (0, 1):
        ctx.set_option("gram_i8", mode)
        fn = (lambda: B.gram_batched(M)) if batch > 1 else (lambda: B.gram(M[0]))
        G = fn()
        G = G if batch > 1 else G[None]
        out[mode] = (t(fn), float((G - ref).abs().max() / ref.abs().max()))
    print("batch %3d n %4d P %6d : f64 %.3f ms (err %.1e)   i8 %.3f ms (err %.1e)   speedup %.2f" % (batch, n, P, out[0][0], out[0][1], out[1][0], out[1][1], out[0][0] / out[1][0]))
ctx.set_option("gram_i8", 0)
