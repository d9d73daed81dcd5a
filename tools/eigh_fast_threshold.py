import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import numpy as np, torch
from vip_amd import backend as B
from eigh_fast_check import gram_with_spectrum, baseline_like
ctx = B.get_context()
def t(fn, reps=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
for n, k in ((450, 20), (500, 20), (513, 20), (540, 20), (580, 20), (620, 20), (700, 20), (800, 30), (513, 40), (600, 50)):
    G = torch.from_numpy(gram_with_spectrum(n, baseline_like(n, seed=n))).cuda()
    ev = torch.zeros(n, dtype=torch.float64, device='cuda'); ec = torch.zeros((n, n), dtype=torch.float64, device='cuda')
    res = {}
    for fast in (1, 0):
        ctx.set_option("eigh_fast", fast); ctx.set_option("eigh_fast_min", 300)
        def call():
            g = G.clone()
            ctx.call("vipmi_eigh_topk_f64", B.ptr(g), 1, n, k, None, B.ptr(ev), B.ptr(ec))
        res[fast] = t(call)
    print("n=%d k=%d fast %.3f ms (reason %d, products %d) exact %.3f ms" % (n, k, res[1], ctx.get_option("eigh_fast_last_reason"), ctx.get_option("eigh_fast_last_products"), res[0]))
