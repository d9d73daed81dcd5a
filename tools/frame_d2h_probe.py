import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca
for n, N, k in ((400, 512, 20), (61, 101, 5)):
    cube, ang = synth_adi(n, N, 0); ct = torch.from_numpy(cube).cuda()
    pin = torch.empty((N, N), dtype=torch.float32).pin_memory()
    def a(): return pca(ct, ang, ncomp=k, verbose=False, check_memory=False).cpu()
    def b():
        pin.copy_(pca(ct, ang, ncomp=k, verbose=False, check_memory=False), non_blocking=True); torch.cuda.current_stream().synchronize(); return pin
    def c():
        fr = pca(ct, ang, ncomp=k, verbose=False, check_memory=False); h = torch.empty((N, N), dtype=torch.float32, pin_memory=True); h.copy_(fr, non_blocking=True); torch.cuda.current_stream().synchronize(); return h.numpy()
    for nm, fn in (("t.cpu()", a), ("fixed pinned buffer", b), ("pinned from torch's host allocator per call", c)):
        for _ in range(60): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): fn()
        torch.cuda.synchronize(); print("%dx%dx%d %-46s %.3f ms" % (n, N, N, nm, (time.perf_counter() - t0) / 100 * 1e3))
