"""All kk leading eigenvalues of T by one wave at once (option eigh_many, second launch of split batches) against one wave per
eigenvalue: accuracy on ragged batches, time of the batch and of C3.   python tools/many_check.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
ctx = B.get_context()
rng = np.random.default_rng(0)
for (batch, n, k) in [(1100, 200, 10), (1100, 200, 2), (1100, 120, 16), (1100, 64, 8), (1100, 150, 5), (1100, 100, 12)]:
    X = rng.standard_normal((batch, n, 3 * n)) * (2.0 ** (-np.arange(3 * n) / 6.0))
    G = X @ X.transpose(0, 2, 1)
    G[5] = np.eye(n) * 3.0; G[6] = 0.0; G[7] = np.diag(np.r_[np.ones(n // 2) * 2.0, np.ones(n - n // 2)])     # degenerate spectra
    nact = rng.integers(1, n + 1, size=batch).astype(np.int32); nact[:8] = (1, 2, 3, 4, n, n, n, n)
    for p in range(batch):
        G[p, nact[p]:, :] = 0; G[p, :, nact[p]:] = 0
    res = {}
    for many in (0, 1):
        ctx.set_option("eigh_many", many); ctx.set_option("timing", 1)
        Gt = torch.from_numpy(G).cuda(); na = torch.from_numpy(nact).cuda()
        ev, E = B.eigh_topk(Gt.clone(), k, nact=na)
        ctx.reset_timers()
        ev, E = B.eigh_topk(Gt.clone(), k, nact=na)
        torch.cuda.synchronize()
        t = ctx.stage_ms("eigh")
        ev = ev.cpu().numpy(); E = E.cpu().numpy(); res[many] = ev
        worst_l, worst_r = 0.0, 0.0
        for p in list(range(8)) + list(range(8, batch, max(1, batch // 24))):
            kk = min(k, nact[p])
            w = np.linalg.eigvalsh(G[p])[::-1]
            sc = max(w[0], 1e-300)
            worst_l = max(worst_l, np.abs(ev[p, :kk] - w[:kk]).max() / sc)
            V = E[p, :kk]
            worst_r = max(worst_r, np.abs(G[p] @ V.T - V.T * ev[p, :kk]).max() / sc, np.abs(V @ V.T - np.eye(kk)).max())
        print("batch %d n %d k %d many %d: %.3f ms  eval err %.1e  residual/orth %.1e" % (batch, n, k, many, t, worst_l, worst_r), flush=True)
    print("   eigenvalues, many vs one wave each: max rel diff %.1e" % (np.abs(res[0][:, :k] - res[1][:, :k]).max() / np.abs(res[0]).max()))
    ctx.set_option("timing", 0)
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca_annular
cube, ang = synth_adi(400, 512, 0); ct = torch.from_numpy(cube).cuda()
f = lambda: pca_annular(ct, ang, asize=32, ncomp=10, fwhm=4, delta_rot=(0.1, 1), verbose=False)
out = {}
for many in (0, 1, 0, 1):
    ctx.set_option("eigh_many", many)
    out[many] = f().cpu().numpy(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); print("C3 many %d: %.2f ms" % (many, (time.perf_counter() - t) / 5 * 1e3), flush=True)
print("C3 frames: max |diff| %.2e (frame scale %.2e)" % (np.nanmax(np.abs(out[0] - out[1])), np.nanmax(np.abs(out[0]))))
