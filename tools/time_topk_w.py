"""n = 600 .. 2000 leading-k solver: matrix-in-L2 kernel (9 vectors in LDS) against the 3-vector kernel built for n > 2048."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
rng = np.random.default_rng(0)
ctx = B.get_context()
for n, k in ((600, 20), (1000, 20), (1500, 30), (2000, 50)):
    X = rng.standard_normal((n, 3 * n)); X[:, :5] *= 10
    Gh = X @ X.T
    w = np.linalg.eigvalsh(Gh)[::-1][:k]
    G = torch.from_numpy(Gh).cuda()[None]
    evals = torch.zeros((1, n), dtype=torch.float64, device="cuda"); evecs = torch.zeros((1, n, n), dtype=torch.float64, device="cuda")
    for xl in (2048, 512):
        ctx.set_option("eigh_xl_min", xl)
        best = 1e9
        for rep in range(3):
            g2 = G.clone(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.call("vipmi_eigh_topk_f64", B.ptr(g2), 1, n, k, 0, B.ptr(evals), B.ptr(evecs))
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        err = np.abs(evals[0, :k].cpu().numpy() - w).max() / w[0]
        print("n=%d k=%d %s: %.3f ms  (eigenvalue error %.1e)" % (n, k, "3-vector kernel" if xl == 512 else "9-vector kernel", best, err))
