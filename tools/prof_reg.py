"""One C3-like batch (3200 problems of 200 x 200, k = 10) through the batched eigensolver, for a counter run: python tools/prof_reg.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
rng = np.random.default_rng(0)
n, k, batch = 200, 10, 3200
X = rng.standard_normal((n, 3 * n)); X[:, :5] *= 10
G = X @ X.T
ctx = B.get_context()
Gt = torch.from_numpy(np.stack([G] * batch)).cuda()
nact = torch.full((batch,), n, dtype=torch.int32, device="cuda")
evals = torch.zeros((batch, n), dtype=torch.float64, device="cuda"); evecs = torch.zeros((batch, n, n), dtype=torch.float64, device="cuda")
for rep in range(2):
    g2 = Gt.clone(); torch.cuda.synchronize()
    ctx.call("vipmi_eigh_topk_f64", B.ptr(g2), batch, n, k, B.ptr(nact), B.ptr(evals), B.ptr(evecs))
torch.cuda.synchronize()
