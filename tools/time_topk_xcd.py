"""The multi-workgroup eigensolver with its workgroups spread over the XCDs (eigh_one_xcd=0) or on one XCD (1), and the
workgroup count: latency of one problem and agreement of the results.   python tools/time_topk_xcd.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
rng = np.random.default_rng(0)
ctx = B.get_context()
ctx.set_option("eigh_fast", 0)
import sys
SIZES = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]] or [(300, 20), (400, 20), (400, 50), (640, 30)]
for n, k in SIZES:
    X = rng.standard_normal((n, 3 * n)); X[:, :5] *= 10
    Gh = X @ X.T
    G = torch.from_numpy(Gh).cuda()[None]
    w = np.linalg.eigvalsh(Gh)[::-1][:k]
    ref = None
    for xcd, W in ((0, 0), (1, 0)) + (((1, 8), (1, 12), (1, 16), (1, 20), (1, 24), (1, 32)) if len(sys.argv) == 1 else ()):
        ctx.set_option("eigh_one_xcd", xcd); ctx.set_option("eigh_w", W)
        evals = torch.zeros((1, n), dtype=torch.float64, device="cuda"); evecs = torch.zeros((1, n, n), dtype=torch.float64, device="cuda")
        best = 1e9
        try:
            for rep in range(5):
                g2 = G.clone(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ctx.call("vipmi_eigh_topk_f64", B.ptr(g2), 1, n, k, 0, B.ptr(evals), B.ptr(evecs))
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
        except Exception as e:
            print("n=%d k=%d one_xcd=%d W=%d: %s" % (n, k, xcd, W, str(e)[:80])); continue
        err = np.abs(evals[0, :k].cpu().numpy() - w).max() / w[0]
        V = evecs[0, :k].cpu().numpy()
        res = np.abs(Gh @ V.T - V.T * evals[0, :k].cpu().numpy()).max() / w[0]
        if ref is None: ref = V.copy()
        print("n=%d k=%d one_xcd=%d W=%2d: %.3f ms  eigenvalue err %.1e  residual %.1e  max|dV| vs first %.1e" % (n, k, xcd, W, best, err, res, np.abs(V - ref).max()))
