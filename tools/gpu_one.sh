#!/bin/bash
timeout 600 python tools/soak.py 600 2>&1 | grep -v amdgpu | tail -3
timeout 600 python - <<'PY'
# soak of the batched / annular paths: 40 annular calls and 40 4-D calls, bit-identical outputs
import sys; sys.path.insert(0, ".")
import numpy as np, torch, time
from vip_amd.psfsub import pca, pca_annular
from vip_amd.synth import synth_adi
cube, ang = synth_adi(200, 256, 1); ct = torch.from_numpy(cube).cuda()
ref = pca_annular(ct, ang, asize=16, ncomp=8, fwhm=4, delta_rot=(0.1, 1), verbose=False)
t = time.perf_counter()
for i in range(40):
    o = pca_annular(ct, ang, asize=16, ncomp=8, fwhm=4, delta_rot=(0.1, 1), verbose=False)
    assert torch.equal(torch.nan_to_num(o), torch.nan_to_num(ref)), i
torch.cuda.synchronize(); print("annular soak ok, %.2f ms per call" % ((time.perf_counter() - t) / 40 * 1e3))
c4 = torch.stack([torch.from_numpy(synth_adi(100, 128, 5 + i)[0]) for i in range(8)]).cuda()
a4 = np.linspace(0, 80, 100)
ref = pca(c4, a4, ncomp=12, verbose=False)
for i in range(40):
    o = pca(c4, a4, ncomp=12, verbose=False)
    assert torch.equal(o, ref), i
print("4-D soak ok")
PY
