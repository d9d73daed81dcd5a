"""Step-loop segments of the multi-workgroup eigensolver (library built with -DVIPMI_TRI_PROFILE, VIPMI_LIB_PATH):
s_memtime ticks of wave 0 of workgroup 0, as fractions of the step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
rng = np.random.default_rng(0)
ctx = B.get_context()
ctx.set_option("eigh_fast", 0)
for o in sys.argv[1:]:
    a, b = o.split("="); ctx.set_option(a, int(b))
for n, k in ((256, 20), (400, 20), (512, 20)):
    X = rng.standard_normal((n, 3 * n)); X[:, :5] *= 10
    G = torch.from_numpy(X @ X.T).cuda()[None]
    evals = torch.zeros((1, n), dtype=torch.float64, device="cuda"); evecs = torch.zeros((1, n, n), dtype=torch.float64, device="cuda")
    for rep in range(3):
        g2 = G.clone(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.call("vipmi_eigh_topk_f64", B.ptr(g2), 1, n, k, 0, B.ptr(evals), B.ptr(evecs))
        e1.record(); torch.cuda.synchronize()
    sg = evals[0, n - 8:n - 3].cpu().numpy()
    tot = sg.sum()
    print("n=%d k=%d: %.3f ms; step loop = %.0f ticks per step: row pass + stores %.0f %% | counter barrier %.0f %% | gather + barrier %.0f %% | K, w, row update + barrier %.0f %% | reflector + barrier %.0f %%" % (
        n, k, e0.elapsed_time(e1), tot / (n - 2), *(100 * sg / tot)))
