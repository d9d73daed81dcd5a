"""Check / time the large-n top-k eigensolver (512 < n <= 2048) against numpy on the GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
rng = np.random.default_rng(0)
for (n, k) in [(513, 5), (600, 20), (1000, 50), (1024, 64), (2000, 50), (2048, 10)]:
    M = rng.standard_normal((n, n + 50)) * (2.0 ** (-np.arange(n + 50) / 40.0))
    G = M @ M.T
    w, E = np.linalg.eigh(G); w = w[::-1]; E = E[:, ::-1]
    ctx = B.get_context(); ctx.set_option("timing", 1)
    for method in (0, 1):
        ctx.set_option("eigh_method", method); ctx.reset_timers()
        try:
            ev, ec = B.eigh_topk(torch.from_numpy(G.copy()).cuda(), k)
        except Exception as exc:
            print("n=%d k=%d method=%d failed: %s" % (n, k, method, exc)); continue
        torch.cuda.synchronize()
        t = ctx.stage_ms("eigh")
        ev = ev.cpu().numpy(); X = ec.cpu().numpy().T
        Pk = X @ X.T; Pr = E[:, :k] @ E[:, :k].T
        print("n=%d k=%d method=%d  %.2f ms  eval relerr %.2e  projector err %.2e  orth %.2e" % (
            n, k, method, t, np.abs(ev - w[:k]).max() / w[0], np.abs(Pk - Pr).max(), np.abs(X.T @ X - np.eye(k)).max()))
