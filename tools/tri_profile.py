"""Phase clocks of the batched top-k eigensolver (library built with -DVIPMI_TRI_PROFILE): s_memtime stamps of problem 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
rng = np.random.default_rng(0)
for n, k, batch in ((200, 10, 1), (200, 10, 400), (200, 10, 1600)):
    X = rng.standard_normal((n, 3 * n)); X[:, :5] *= 10
    G = X @ X.T
    for reg in (1, 0):
        ctx = B.get_context(); ctx.set_option("eigh_reg", reg)
        Gt = torch.from_numpy(np.stack([G] * batch)).cuda()
        nact = torch.full((batch,), n, dtype=torch.int32, device="cuda")
        evals = torch.zeros((batch, n), dtype=torch.float64, device="cuda"); evecs = torch.zeros((batch, n, n), dtype=torch.float64, device="cuda")
        for rep in range(2):
            g2 = Gt.clone()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.call("vipmi_eigh_topk_f64", B.ptr(g2), batch, n, k, B.ptr(nact), B.ptr(evals), B.ptr(evecs))
            e1.record(); torch.cuda.synchronize()
        st = evals[0, n - 8:n - 2].cpu().numpy()
        d = np.diff(st) / 100.0      # s_memtime ticks at 100 MHz -> us
        sg = evals[0, n - 16:n - 11].cpu().numpy()
        print("   step-loop segments of wave 0 (cycles/step): update+corner %.0f | barrier %.0f | vector phases (p, K, w, norms) %.0f | reflector %.0f | (unused %.0f)" % tuple(sg / (n - 2)))
        print("n=%d k=%d batch=%d reg=%d: %.3f ms | us: tridiag %.0f  eigenvalues %.0f  inverse-iteration %.0f  gram-schmidt %.0f  back-transform %.0f" % (
            n, k, batch, reg, e0.elapsed_time(e1), *d))
