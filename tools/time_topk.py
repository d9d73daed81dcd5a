"""Latency of ONE top-k eigenproblem (vipmi_eigh_topk_f64) at the sizes of the full-frame path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
rng = np.random.default_rng(0)
ctx = B.get_context()
for n, k in ((100, 10), (200, 10), (400, 20), (400, 50), (1000, 20), (2000, 50)):
    X = rng.standard_normal((n, 3 * n)); X[:, :5] *= 10
    G = torch.from_numpy(X @ X.T).cuda()[None]
    nact = torch.full((1,), n, dtype=torch.int32, device="cuda")
    evals = torch.zeros((1, n), dtype=torch.float64, device="cuda"); evecs = torch.zeros((1, n, n), dtype=torch.float64, device="cuda")
    best = 1e9
    for rep in range(4):
        g2 = G.clone(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.call("vipmi_eigh_topk_f64", B.ptr(g2), 1, n, k, 0, B.ptr(evals), B.ptr(evecs))
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    w = np.linalg.eigvalsh(X @ X.T)[::-1][:k]
    err = np.abs(evals[0, :k].cpu().numpy() - w).max() / w[0]
    print("n=%d k=%d: %.3f ms  (eigenvalue error %.1e of the largest)" % (n, k, best, err))
