"""Many eigenvectors of one Gram matrix (pca(ncomp > 64), CEVR): the matrix-in-L2 tridiagonal solver against the one-sided Jacobi kernel."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi_device
ctx = B.get_context()
for n in (200, 400, 512, 800):
    ct, _ = synth_adi_device(n, 128, seed=n); G = B.gram(ct.reshape(n, -1))
    for k in (65, 100, 200, n):
        for meth in (0, 1):
            ctx.set_option("eigh_method", meth)
            try:
                B.eigh_topk(G.clone(), k); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(2): ev, ec = B.eigh_topk(G.clone(), k)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2 * 1e3
                X = ec.cpu().numpy().T; Gh = G.cpu().numpy()
                res = np.abs(Gh @ X - X * ev.cpu().numpy()[None]).max() / ev[0].item()
                print("n=%d k=%d method=%d: %.2f ms residual %.1e sweeps %d" % (n, k, meth, dt, res, ctx.get_option("eigh_last_sweeps")), flush=True)
            except Exception as e:
                print("n=%d k=%d method=%d: %s" % (n, k, meth, str(e)[:80]), flush=True)
ctx.set_option("eigh_method", 0)
