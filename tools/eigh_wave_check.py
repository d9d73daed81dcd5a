"""The lone eigenproblem of 129 .. 448 rows: wave-resident tridiagonalisation (eigh_wave.hip, option eigh_wave) against the
LDS-resident multi-workgroup kernel and numpy.   python tools/eigh_wave_check.py [n k ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi_device
args = [int(a) for a in sys.argv[1:]]
cases = list(zip(args[0::2], args[1::2])) or [(400, 20), (448, 20), (385, 5), (320, 40), (257, 10), (200, 64), (129, 3)]
ctx = B.get_context()
for n, k in cases:
    ct, ang = synth_adi_device(n, 256, seed=n)
    M = ct.reshape(n, -1)
    M = M - M.mean(0, keepdim=True)
    G = B.gram(M)[None].clone()
    Gh = G[0].cpu().numpy()
    w, E = np.linalg.eigh(Gh); w = w[::-1]; E = E[:, ::-1]
    evals = torch.zeros((1, n), dtype=torch.float64, device="cuda"); evecs = torch.zeros((1, n, n), dtype=torch.float64, device="cuda")
    for wave in (0, 1):
        ctx.set_option("eigh_wave", wave)
        ts = []
        for rep in range(6):
            g2 = G.clone(); evals.zero_(); evecs.zero_(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ctx.call("vipmi_eigh_topk_f64", B.ptr(g2), 1, n, k, 0, B.ptr(evals), B.ptr(evecs)); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ev = evals[0, :k].cpu().numpy(); X = evecs[0, :k].cpu().numpy().T
        res = np.abs(Gh @ X - X * ev[None, :]).max() / w[0]
        Pk = X @ X.T; Pr = E[:, :k] @ E[:, :k].T
        print("n=%d k=%d eigh_wave=%d: min %.3f ms median %.3f | eval relerr %.1e residual %.1e projector %.1e orth %.1e" % (
            n, k, wave, min(ts), sorted(ts)[len(ts) // 2], np.abs(ev - w[:k]).max() / w[0], res, np.abs(Pk - Pr).max(),
            np.abs(X.T @ X - np.eye(k)).max()), flush=True)
B.check_deferred() if hasattr(B, "check_deferred") else None
