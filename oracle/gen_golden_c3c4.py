"""Full-size pins of BASELINE.json configs[2] and configs[3] against the REAL reference (imported read-only through
oracle/_shim.py); each takes tens of minutes of CPU and runs only in the build container:

    python oracle/gen_golden_c3c4.py c3     # pca_annular(400 x 512 x 512, 8 annuli, ncomp=10)
    python oracle/gen_golden_c3c4.py c4     # pca(39 x 200 x 256 x 256, ncomp=20), per-channel PCA + spectral mean

Outputs only are stored (the cubes are regenerated from their seeds at test time).
"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _shim, ref_cpu as O  # noqa: E402

warnings.simplefilter("ignore")
ref = _shim.load()
which = sys.argv[1]
t0 = time.time()
if which == "c3":
    cube, ang = O.synth_adi(400, 512, seed=0)
    co, cd, fr = ref.pca_annular(cube, ang, ncomp=10, asize=32, fwhm=4, delta_rot=(0.1, 1), n_segments=1,
                                 full_output=True, verbose=False, nproc=1)
    keep = [0, 133, 266, 399]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g16_annular_c3.npz"), frame=fr.astype(np.float32),
                        keep=np.array(keep), out_keep=co[keep].astype(np.float32),
                        out_rowsum=co.reshape(400, -1).astype(np.float64).sum(axis=1))
elif which == "c4":
    c4 = np.stack([O.synth_adi(200, 256, seed=s)[0] for s in range(39)])
    ang = np.linspace(0, 90, 200)
    fo = ref.pca(c4, ang, ncomp=20, full_output=True, verbose=False, nproc=1)
    frame, ifs = fo[0], fo[-1]
    chs = np.array([0, 5, 12, 19, 26, 33, 38])           # per-channel frames kept (the whole stack is 10 MB)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g17_pca4d_c4.npz"), frame=np.asarray(frame, np.float32),
                        ifs_channels=chs, ifs=np.asarray(ifs, np.float32)[chs])
print("%s: %.0f s" % (which, time.time() - t0))
