"""Stage stamps of stages 2-5 of tri_multi_kernel, workgroup 0 (library built with -DVIPMI_TRI_PROFILE: make -C vip_amd/csrc prof;
VIPMI_LIB_PATH=vip_amd/libvipmi_prof.so).   python tools/tri_stage_profile.py [n k] [opt=val ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
rng = np.random.default_rng(0)
ctx = B.get_context()
ctx.set_option("eigh_fast", 0)
nums = [int(a) for a in sys.argv[1:] if "=" not in a]
for o in [a for a in sys.argv[1:] if "=" in a]:
    a, b = o.split("="); ctx.set_option(a, int(b))
n, k = (nums + [400, 20])[:2]
X = rng.standard_normal((n, 3 * n)); X[:, :5] *= 10
G = torch.from_numpy(X @ X.T).cuda()[None]
evals = torch.zeros((1, n), dtype=torch.float64, device="cuda"); evecs = torch.zeros((1, n, n), dtype=torch.float64, device="cuda")
for rep in range(3):
    g2 = G.clone(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx.call("vipmi_eigh_topk_f64", B.ptr(g2), 1, n, k, 0, B.ptr(evals), B.ptr(evecs))
    e1.record(); torch.cuda.synchronize()
st = evals[0, n - 16:n - 10].cpu().numpy()
d = np.diff(st)
if ctx.get_option("eigh_wave") != 0:
    print("n=%d k=%d: %.3f ms; tri_vec_kernel, workgroup 0, s_memtime ticks: multisection %.0f | inverse iteration %.0f | back-transformation %.0f" % (
        n, k, e0.elapsed_time(e1), d[0], d[1], d[2]))
    print("   inverse iteration: LU + first forward pass until %.0f after its start, second forward pass ends at %.0f (of %.0f)" % (
        st[4] - st[1], st[5] - st[1], st[2] - st[1]))
    sys.exit(0)
print("n=%d k=%d: %.3f ms; s_memtime ticks (100 MHz) from stage 2 on: multisection %.0f | inverse iteration %.0f | back-transformation %.0f | barrier %.0f | Gram-Schmidt + output %.0f | total %.0f" % (
    n, k, e0.elapsed_time(e1), *d, st[-1]))
