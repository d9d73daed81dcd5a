import sys; sys.path.insert(0, ".")
import torch
from vip_amd import backend as B
b, n, P = 39, 200, 65536
M = torch.randn(b, n, P, device="cuda")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
print("batched  %.3f ms" % t(lambda: B.gram_batched(M)))
print("loop     %.3f ms" % t(lambda: [B.gram(M[i]) for i in range(b)]))
