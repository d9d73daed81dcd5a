"""Batched top-k eigensolver (one workgroup per problem) with 256 / 512 / 1024 threads per workgroup: accuracy vs numpy
and time for an annular-PCA-like batch (400 problems of 200 x 200, k = 10) plus ragged active sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B

rng = np.random.default_rng(0)
for (batch, n, k) in [(400, 200, 10), (400, 200, 30), (600, 120, 16), (300, 333, 8), (20, 200, 10)]:
    X = rng.standard_normal((batch, n, 3 * n)) * (2.0 ** (-np.arange(3 * n) / 6.0))
    G = X @ X.transpose(0, 2, 1)
    nact = rng.integers(max(k, n // 2), n + 1, size=batch).astype(np.int32)
    for p in range(batch):
        G[p, nact[p]:, :] = 0; G[p, :, nact[p]:] = 0
    for nt in (1024, 512, 256):
        ctx = B.get_context(); ctx.set_option("eigh_nt", nt); ctx.set_option("timing", 1)
        Gt = torch.from_numpy(G).cuda(); na = torch.from_numpy(nact).cuda()
        ev, E = B.eigh_topk(Gt.clone(), k, nact=na)
        ctx.reset_timers()
        ev, E = B.eigh_topk(Gt.clone(), k, nact=na)
        torch.cuda.synchronize()
        t = ctx.stage_ms("eigh")
        ev = ev.cpu().numpy(); E = E.cpu().numpy()
        worst_l, worst_r = 0.0, 0.0
        for p in range(0, batch, max(1, batch // 16)):
            w = np.linalg.eigvalsh(G[p])[::-1]
            worst_l = max(worst_l, np.abs(ev[p, :k] - w[:k]).max() / w[0])
            V = E[p, :k]
            worst_r = max(worst_r, np.abs(G[p] @ V.T - V.T * ev[p, :k]).max() / w[0], np.abs(V @ V.T - np.eye(k)).max())
        print("batch %d n %d k %d nt %d: %.3f ms  eval err %.1e  residual/orth %.1e" % (batch, n, k, nt, t, worst_l, worst_r))
    ctx.set_option("eigh_nt", 0)
