"""Check / time the tridiagonal top-k eigensolver against numpy on the GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi

def run(n, N, k, method):
    cube, ang = synth_adi(n, N, seed=0)
    ct = torch.from_numpy(cube).cuda()
    ctx = B.get_context()
    ctx.set_option("eigh_method", method)
    ctx.set_option("timing", 1)
    M = ct.reshape(n, -1)
    out = B.pca_project(M, k, want_recon=False, want_pcs=True)
    torch.cuda.synchronize()
    ctx.reset_timers()
    out = B.pca_project(M, k, want_recon=False, want_pcs=True)
    torch.cuda.synchronize()
    t = ctx.stage_ms("eigh")
    return out, t

for (n, N, k) in [(400, 128, 20), (400, 512, 20), (200, 256, 20), (50, 128, 5), (100, 101, 64), (512, 64, 30), (7, 32, 7), (3, 16, 2)]:
    o1, t1 = run(n, N, k, 1)
    o2, t2 = run(n, N, k, 0)
    r1 = o1[0].cpu().numpy(); r2 = o2[0].cpu().numpy()
    print("n=%d N=%d k=%d  jacobi %.3f ms  tri %.3f ms  max|res diff| %.3e  (max|res| %.2f)" % (n, N, k, t1, t2, np.abs(r1 - r2).max(), np.abs(r1).max()))

print("direct top-k API vs numpy")
for (n, N, k, multi) in [(400, 128, 20, 0), (400, 128, 20, 1), (200, 128, 20, 1), (128, 64, 10, 1), (512, 64, 64, 1), (97, 64, 33, 1)]:
    cube, ang = synth_adi(n, N, seed=1)
    M = cube.reshape(n, -1).astype(np.float64)
    G = M @ M.T
    w, E = np.linalg.eigh(G); w = w[::-1]; E = E[:, ::-1]
    Gt = torch.from_numpy(G).cuda()
    ctx = B.get_context(); ctx.set_option("eigh_method", 0); ctx.set_option("eigh_multi", multi); ctx.set_option("timing", 1); ctx.reset_timers()
    evals = torch.zeros((1, n), dtype=torch.float64, device="cuda"); evecs = torch.zeros((1, n, n), dtype=torch.float64, device="cuda")
    ctx.call("vipmi_eigh_topk_f64", B.ptr(Gt), 1, n, k, None, B.ptr(evals), B.ptr(evecs))
    torch.cuda.synchronize()
    ev = evals[0].cpu().numpy(); X = evecs[0, :k].cpu().numpy().T
    Pk = X @ X.T; Pr = E[:, :k] @ E[:, :k].T
    print("multi", multi, "eigh ms %.3f" % ctx.stage_ms("eigh"))
    print("n=%d k=%d  eval relerr %.2e  projector err %.2e  orth %.2e" % (
        n, k, np.abs(ev[:k] - w[:k]).max() / w[0], np.abs(Pk - Pr).max(), np.abs(X.T @ X - np.eye(k)).max()))
