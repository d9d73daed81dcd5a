"""Time the Gram kernel alone:  python tools/time_gram.py n N [opt=val ...]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vip_amd import backend as B
n, N = int(sys.argv[1]), int(sys.argv[2])
ctx = B.get_context()
for kv in sys.argv[3:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
M = torch.randn(n, N * N, device="cuda")
for _ in range(2): B.gram(M)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): G = B.gram(M)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(n, N, sys.argv[3:], "%.3f ms" % ms, "%.1f TF/s (n^2 P flops)" % (n * n * N * N / ms / 1e9))
