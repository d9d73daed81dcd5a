"""float64 route at C2 size: the centring kernel alone and the whole resident call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
ctx = B.get_context()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
n, N = 400, 512
c64 = (7000 + 45 * torch.randn(n, N, N, device="cuda", dtype=torch.float64))
D = torch.empty(n, N * N, device="cuda"); mu = torch.empty(N * N, device="cuda", dtype=torch.float64)
for mode in (1, 2):
    print("center_f64 mode %d: %.3f ms" % (mode, t(lambda: ctx.call("vipmi_center_f64", B.ptr(c64), n, N * N, mode, B.ptr(D), B.ptr(mu), None))))
ang = np.linspace(0, 90, n)
print("pca(float64 cuda cube, k = 20): %.3f ms" % t(lambda: pca(c64, ang, ncomp=20, verbose=False, check_memory=False)))
print("pca(float32 cuda cube, k = 20): %.3f ms" % t(lambda: pca(c64.float(), ang, ncomp=20, verbose=False, check_memory=False)))
