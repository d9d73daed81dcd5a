"""The lone n = 400 (C2) eigenproblem: workgroup count / one-XCD layout.   python tools/time_eigh400.py [n k]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi_device
n, k = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (400, 20)
ct, ang = synth_adi_device(n, 512 if n <= 400 else 256, seed=0)
M = ct.reshape(n, -1)
M = M - M.mean(0, keepdim=True)
G = B.gram(M)[None].clone()
w = np.linalg.eigvalsh(G[0].cpu().numpy())[::-1][:k]
ctx = B.get_context()
evals = torch.zeros((1, n), dtype=torch.float64, device="cuda"); evecs = torch.zeros((1, n, n), dtype=torch.float64, device="cuda")
ref = None
for xcd, W in ((-1, 0), (0, 16), (1, 16), (1, 20), (1, 24), (1, 28), (1, 32), (0, 32)):
    ctx.set_option("eigh_one_xcd", xcd); ctx.set_option("eigh_w", W)
    ts = []
    for rep in range(6):
        g2 = G.clone(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.call("vipmi_eigh_topk_f64", B.ptr(g2), 1, n, k, 0, B.ptr(evals), B.ptr(evecs)); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    err = np.abs(evals[0, :k].cpu().numpy() - w).max() / w[0]
    v = evecs[0, :k].clone()
    if ref is None: ref = v
    print("n=%d k=%d one_xcd=%2d W=%2d: min %.3f ms median %.3f  (eigenvalue error %.1e, vectors identical to default: %s)" % (
        n, k, xcd, W, min(ts), sorted(ts)[len(ts) // 2], err, bool(torch.equal(v, ref))))
