"""Full-size pin of BASELINE.json configs[1] against the REAL reference (imported read-only through oracle/_shim.py):
pca(400 x 512 x 512 synthetic ADI cube, ncomp=20, svd_mode='lapack', nproc=1).  Takes ~15 min of CPU (the reference's
per-frame FFT derotation); runs only in the build container:

    python oracle/gen_golden_c2.py

Stores outputs only (the cube is regenerated from its seed at test time): the final frame, the derotated residuals of
four frames and a checksum of the residual cube.
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
cube, ang = O.synth_adi(400, 512, seed=0)
t0 = time.time()
frame, pcs, recon, res, resd = ref.pca(cube, ang, ncomp=20, svd_mode="lapack", full_output=True, verbose=False, nproc=1)
print("reference pca at C2: %.0f s" % (time.time() - t0))
keep = [0, 133, 266, 399]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g15_pca_c2.npz"), seed=0, frame=frame.astype(np.float32),
                    frame_dtype=str(frame.dtype), keep=np.array(keep), res_keep=res[keep].astype(np.float32),
                    resd_keep=resd[keep].astype(np.float32),
                    res_rowsum=res.reshape(400, -1).astype(np.float64).sum(axis=1),
                    res_abs_max=float(np.abs(res).max()))
print("saved")
