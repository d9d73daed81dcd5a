"""Batched top-k eigensolver: one launch against tridiagonalisation + rest as two launches (option eigh_split)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
rng = np.random.default_rng(0)
ctx = B.get_context()
for (batch, n, k) in [(400, 200, 10), (1600, 200, 10), (400, 120, 5), (256, 200, 16), (3200, 200, 10)]:
    X = rng.standard_normal((min(batch, 400), n, 3 * n)) * (2.0 ** (-np.arange(3 * n) / 6.0))
    G = X @ X.transpose(0, 2, 1)
    G = np.concatenate([G] * (batch // G.shape[0]))
    nact = rng.integers(max(k, n // 2), n + 1, size=batch).astype(np.int32)
    for p in range(batch):
        G[p, nact[p]:, :] = 0; G[p, :, nact[p]:] = 0
    Gt = torch.from_numpy(G).cuda(); na = torch.from_numpy(nact).cuda()
    res = {}
    for sp in (0, 1):
        ctx.set_option("eigh_split", sp)
        best = 1e9
        for rep in range(3):
            g2 = Gt.clone(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ev, E = B.eigh_topk(g2, k, nact=na); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        res[sp] = (best, ev.cpu().numpy(), E.cpu().numpy())
    ctx.set_option("eigh_split", -1)
    d_ev = np.abs(res[0][1][:, :k] - res[1][1][:, :k]).max() / np.abs(res[0][1]).max()
    d_E = np.abs(np.abs(res[0][2][:, :k]) - np.abs(res[1][2][:, :k])).max()
    print("batch %d n %d k %d: one launch %.3f ms, split %.3f ms; difference eigenvalues %.1e vectors %.1e" % (batch, n, k, res[0][0], res[1][0], d_ev, d_E))
