import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, warnings
from vip_amd import backend as B
warnings.simplefilter("ignore")
for n in (1, 2, 3, 5, 7, 8, 33, 63, 64, 65, 100, 128, 129, 400, 449, 1000):
    for P in (1, 4, 32, 100, 256):
        rng = np.random.default_rng(n * 1000 + P)
        cube = rng.standard_normal((n, P)).astype(np.float32)
        got = B.collapse(torch.from_numpy(cube).cuda().reshape(n, P, 1), "median").cpu().numpy().reshape(P)
        exp = np.nanmedian(cube, axis=0)
        bad = np.nonzero(~((got == exp) | (np.isnan(got) & np.isnan(exp))))[0]
        if bad.size:
            print("n=%d P=%d: %d bad, first px %d got %r exp %r" % (n, P, bad.size, bad[0], got[bad[0]], exp[bad[0]]))
print("done")
