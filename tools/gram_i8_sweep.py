import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vip_amd import backend as B
from vip_amd.synth import synth_adi_device
n, N = (int(a) for a in sys.argv[1:3]) if len(sys.argv) > 2 else (400, 512)
SL = [int(a) for a in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0, 24, 32, 48, 64, 96]
ct, ang = synth_adi_device(n, N, seed=0)
M = ct.reshape(n, -1)
ctx = B.get_context()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
ctx.set_option("gram_i8", 0); print("f64 mfma      %.3f ms" % t(lambda: B.gram(M)))
ctx.set_option("gram_i8", 1)
for nbuf in (1, 2):
    for sl in SL:
        ctx.set_option("gram_i8_nbuf", nbuf); ctx.set_option("gram_i8_slices", sl)
        print("i8 nbuf=%d slices=%d  %.3f ms" % (nbuf, sl, t(lambda: B.gram(M))))
