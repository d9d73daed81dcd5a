"""Per-phase s_memtime ticks (100 MHz) of the batched register-resident eigensolver, problem 0 of a C3-like batch
(3200 problems of 200 x 200, k = 10), library built with -DVIPMI_TRI_PROFILE (make -C vip_amd/csrc prof;
VIPMI_LIB_PATH=vip_amd/libvipmi_prof.so).   python tools/tri_batch_profile.py [batch n k]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
nums = [int(a) for a in sys.argv[1:] if "=" not in a]
batch, n, k = (nums + [3200, 200, 10])[:3]
ctx = B.get_context()
for o in [a for a in sys.argv[1:] if "=" in a]:
    a, b = o.split("="); ctx.set_option(a, int(b))
rng = np.random.default_rng(0)
X = rng.standard_normal((n, 3 * n)); X[:, :5] *= 10
G0 = torch.from_numpy(X @ X.T).cuda()
evals = torch.zeros((batch, n), dtype=torch.float64, device="cuda"); evecs = torch.zeros((batch, n, n), dtype=torch.float64, device="cuda")
for rep in range(3):
    G = G0[None].repeat(batch, 1, 1).contiguous(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx.call("vipmi_eigh_topk_f64", B.ptr(G), batch, n, k, 0, B.ptr(evals), B.ptr(evecs))
    e1.record(); torch.cuda.synchronize()
ev = evals[0].cpu().numpy()
seg = ev[n - 16:n - 11]
st = ev[n - 8:n - 2]
print("batch %d of %d x %d, k = %d: %.3f ms" % (batch, n, n, k, e0.elapsed_time(e1)))
print("launch 1 (tridiagonalisation), problem 0, shader cycles summed over the %d steps: update+corner %.0f | barrier wait %.0f | vector phases %.0f | reflector+barriers %.0f  (sum %.0f= %.0f per step)" % (
    n - 2, seg[0], seg[1], seg[2], seg[3], seg[:4].sum(), seg[:4].sum() / (n - 2)))
d = np.diff(st)
print("launch 2, problem 0, shader cycles between stamps: start->T %.0f | multisection %.0f | inverse iteration %.0f | Gram-Schmidt %.0f | back-transformation+output %.0f  (total %.0f)" % (
    d[0], d[1], d[2], d[3], d[4], st[5] - st[0]))
ls = ev[n - 32:n - 25]
if ls[0] > 0:
    dl = np.diff(ls)
    print("   inverse iteration (active wave, lane 0): LU + first right-hand side %.0f | wait for the stores %.0f | back 1 %.0f | forward 2 %.0f | back 2 %.0f  (after stamp 2: starts at %.0f)" % (
        dl[0], dl[1], dl[3 - 1] + 0 * dl[2], dl[3], dl[5], ls[0] - st[2]))
    print("   raw:", [int(x) for x in dl])
