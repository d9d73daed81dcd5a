"""The feature combinations of tests/test_gpu_fuzz.py::test_pca_feature_combinations_random at MEDIUM sizes (40..200 frames of
80..220 px): RDI, cube_sig, weighted mean, temporal modes, grids of PCs, 4-D cubes, median subtraction, ADI+mSDI single / double.
`gen DIR first count` writes the oracle's results (any host), `check DIR` compares the device results on the GPU box."""
import sys, os, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import ref_cpu as O

mode, refdir = sys.argv[1], sys.argv[2]
os.makedirs(refdir, exist_ok=True)
if mode == "gen":
    seeds = range(int(sys.argv[3]), int(sys.argv[3]) + int(sys.argv[4]))
else:
    seeds = sorted(int(f[4:-4]) for f in os.listdir(refdir) if f.startswith("ref_") and f.endswith(".npz"))
    from vip_amd.psfsub import pca, median_sub
SCALINGS = (None, "temp-mean", "spat-mean", "temp-standard", "spat-standard")
TOL = 1e-4
bad = 0


def cube_of(rng, n, N):
    return O.synth_adi(n, N, seed=int(rng.integers(1 << 30)))[0].astype(np.float32)


for seed in seeds:
    rng = np.random.default_rng(43000 + seed)
    n = int(rng.integers(40, 200)); N = int(rng.integers(80, 220))
    feat = seed % 8
    if feat in (5, 7):
        n = int(rng.integers(12, 60)); N = int(rng.integers(60, 140))
    cube = cube_of(rng, n, N)
    ang = np.linspace(0, float(rng.uniform(40, 170)), n)
    k = int(rng.integers(1, 12))
    scaling = SCALINGS[rng.integers(len(SCALINGS))]
    t0 = time.time()
    calls = []          # (label, device thunk, oracle thunk, tolerance)
    if feat == 0:
        refc = cube_of(rng, int(rng.integers(k + 1, 120)), N)
        calls.append(("rdi", lambda: pca(cube, ang, ncomp=k, cube_ref=refc, scaling=scaling, verbose=False),
                      lambda: O.pca_fullframe(cube, ang, ncomp=k, cube_ref=refc, scaling=scaling), TOL))
    elif feat == 1:
        sig = (0.1 * np.abs(cube_of(rng, n, N))).astype(np.float32)
        calls.append(("cube_sig", lambda: pca(cube, ang, ncomp=k, cube_sig=sig, scaling=scaling, verbose=False),
                      lambda: O.pca_fullframe(cube, ang, ncomp=k, cube_sig=sig, scaling=scaling), TOL))
    elif feat == 2:
        w = rng.uniform(0.1, 2.0, n)
        calls.append(("wmean", lambda: pca(cube, ang, ncomp=k, weights=w, collapse="wmean", verbose=False),
                      lambda: O.pca_fullframe(cube, ang, ncomp=k, weights=w, collapse="wmean"), TOL))
    elif feat == 3:
        calls.append(("left_eigv", lambda: pca(cube, ang, ncomp=k, left_eigv=True, verbose=False),
                      lambda: O.pca_fullframe(cube, ang, ncomp=k, left_eigv=True), TOL))
    elif feat == 4:
        rng_pcs = (1, min(n, 15), 3)
        calls.append(("grid", lambda: pca(cube, ang, ncomp=rng_pcs, scaling=scaling, full_output=True, verbose=False)[0],
                      lambda: O.pca_grid_frames(cube, ang, rng_pcs, scaling=scaling, full_output=True)[0], TOL))
    elif feat == 5:
        nch = int(rng.integers(2, 6))
        c4 = np.stack([cube_of(rng, n, N) for _ in range(nch)])
        cifs = ("mean", "median")[rng.integers(2)]
        calls.append(("4d", lambda: pca(c4, ang, ncomp=min(k, 6), scaling=scaling, collapse_ifs=cifs, verbose=False),
                      lambda: O.pca_4d(c4, ang, ncomp=min(k, 6), scaling=scaling, collapse_ifs=cifs), TOL))
    elif feat == 6:
        calls.append(("medsub", lambda: median_sub(cube, ang, verbose=False), lambda: O.median_sub_fullfr(cube, ang), 5e-5))
    else:
        nch = int(rng.integers(2, 5)); n4 = min(n, 16)
        c4 = np.stack([cube_of(rng, n4, N) for _ in range(nch)])
        scal = np.linspace(float(rng.uniform(1.05, 1.35)), 1.0, nch)
        a4 = ang[:n4]
        calls.append(("msdi double", lambda: pca(c4, a4, scale_list=scal, ncomp=(2, 3), adimsdi="double", verbose=False),
                      lambda: O.pca_adimsdi_double(c4, a4, scal, (2, 3)), 5e-4))
        calls.append(("msdi single", lambda: pca(c4, a4, scale_list=scal, ncomp=3, adimsdi="single", verbose=False),
                      lambda: O.pca_adimsdi_single(c4, a4, scal, 3), 5e-4))
    path = os.path.join(refdir, "ref_%d.npz" % seed)
    try:
        if mode == "gen":
            np.savez_compressed(path, **{"r%d" % i: np.asarray(c[2]()) for i, c in enumerate(calls)})
            print("gen  seed %d feature %d n %d N %d  (%.0f s)" % (seed, feat, n, N, time.time() - t0), flush=True)
            continue
        refs = np.load(path)
        for i, (label, dev, _o, tol) in enumerate(calls):
            a, b = np.asarray(dev()), refs["r%d" % i]
            ok = np.isfinite(b)
            assert a.shape == b.shape and np.array_equal(np.isfinite(a), ok), "%s: shape / NaN pattern" % label
            d = float(np.abs(a[ok] - b[ok]).max())
            assert d < tol * max(1.0, float(np.abs(b[ok]).max()) / 10.0), "%s: max|d| %.2e" % (label, d)
            print("ok   seed %d %s n %d N %d k %d %s: %.2e" % (seed, label, n, N, k, scaling, d), flush=True)
    except Exception as e:
        bad += 1
        print("FAIL seed %d feature %d n %d N %d k %d %s: %s" % (seed, feat, n, N, k, scaling, "".join(traceback.format_exception_only(type(e), e)).strip()[:400]), flush=True)
print("failures:", bad)
