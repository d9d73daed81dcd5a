"""int8 Gram with the operands staged through registers (gram_i8_dma=0) or by LDS-DMA (1): time and bit-equality."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vip_amd import backend as B
from vip_amd.synth import synth_adi_device
ctx = B.get_context()
def t(fn, reps=7):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
for n, N in ((400, 512), (1000, 512), (2000, 1024), (300, 256)):
    ct, ang = synth_adi_device(n, N, seed=0)
    M = ct.reshape(n, -1)
    ctx.set_option("gram_i8", 1)
    out = {}
    for dma in (0, 1):
        ctx.set_option("gram_i8_dma", dma)
        G = B.gram(M).clone()
        out[dma] = (t(lambda: B.gram(M)), G)
    print("n %4d N %4d: registers %.3f ms   dma %.3f ms   bit-equal %s" % (n, N, out[0][0], out[1][0], bool(torch.equal(out[0][1], out[1][1]))))
    del ct, M
ctx.set_option("gram_i8", -1)
