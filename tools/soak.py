"""Soak test: many pipelined pca() calls on two streams; every output frame must be bit-identical to the first one."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cube, ang = synth_adi(400, 512, 0)
ct = torch.from_numpy(cube).cuda()
ref = pca(ct, ang, ncomp=20, verbose=False, check_memory=False).clone()
streams = [torch.cuda.Stream() for _ in range(2)]
B.set_async(True)
outs = []
t0 = time.perf_counter()
for i in range(steps):
    with torch.cuda.stream(streams[i % 2]):
        outs.append(pca(ct, ang, ncomp=20, verbose=False, check_memory=False))
    if len(outs) >= 50:
        torch.cuda.synchronize()
        bad = [j for j, o in enumerate(outs) if not torch.equal(o, ref)]
        assert not bad, bad
        outs = []
torch.cuda.synchronize(); B.check_deferred(); B.set_async(False)
assert all(torch.equal(o, ref) for o in outs)
print("soak: %d pipelined calls, all frames bit-identical, %.2f ms per call" % (steps, (time.perf_counter() - t0) / steps * 1e3))
