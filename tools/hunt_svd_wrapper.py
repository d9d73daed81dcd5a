"""svd_wrapper at random shapes (n 2 .. 700 frames, P from below n to 40 k -- wide AND tall matrices --, ncomp up to min(n, P)), every
mode name, V alone / (U, S, V) / left_eigv, numpy float32 / float64 and cuda input: singular values, orthonormality, the projector of
the leading subspace and the reconstruction against numpy's SVD.   python tools/hunt_svd_wrapper.py [first [count]]"""
import sys, os, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd.psfsub.svd import svd_wrapper, SVD_MODES

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
modes = sorted(SVD_MODES)
bad = 0
t00 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(23000 + seed)
    n = int(rng.integers(2, 700)) if seed % 3 else int(rng.integers(2, 60))
    P = int(rng.integers(max(2, n // 3), 40000)) if seed % 4 else int(rng.integers(2, 2 * n + 2))
    if n * P > 2e7:
        P = int(2e7 // n)
    k = int(rng.integers(1, min(n, P, 80) + 1))
    mode = modes[rng.integers(len(modes))]
    X = rng.standard_normal((n, P))
    r = min(n, P, 6)
    X += 8 * rng.standard_normal((n, r)) @ rng.standard_normal((r, P))
    dt = (np.float32, np.float64)[rng.integers(2)]
    M = X.astype(dt)
    variant = ("V", "USV", "left")[rng.integers(3)]
    if variant == "left" and mode in ("eigen", "eigencupy", "eigenpytorch"):
        variant = "V"
    what = "n %d P %d k %d mode %s %s %s" % (n, P, k, mode, np.dtype(dt).name, variant)
    try:
        U0, S0, V0 = np.linalg.svd(M.astype(np.float64), full_matrices=False)
        gap = (S0[k - 1] ** 2 - (S0[k] ** 2 if k < len(S0) else 0.0)) / S0[0] ** 2
        arg = torch.from_numpy(M).cuda() if rng.integers(3) == 0 and dt == np.float32 else M
        out = svd_wrapper(arg, mode, k, False, full_output=(variant == "USV"), left_eigv=(variant == "left"))
        conv = lambda a: (a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)).astype(np.float64)
        if variant == "V":
            V = conv(out)
            assert V.shape == (k, P), "V shape %s" % (V.shape,)
            assert np.abs(V @ V.T - np.eye(k)).max() < 2e-5, "V rows not orthonormal: %.2e" % np.abs(V @ V.T - np.eye(k)).max()
            if gap > 1e-3:
                d = np.abs(V.T @ V - V0[:k].T @ V0[:k]).max() if P <= 3000 else np.abs((V @ V0[:k].T) @ (V0[:k] @ V.T) - np.eye(k)).max()
                assert d < 5e-4 / min(1.0, gap * 50), "leading subspace: %.2e (gap %.1e)" % (d, gap)
        elif variant == "USV":
            U, S, V = [conv(a) for a in out]
            assert V.shape == (k, P), "V shape %s" % (V.shape,)
            Sk = S[:k] if mode not in ("eigen", "eigencupy", "eigenpytorch") else S[:k]
            assert np.abs(Sk - S0[:k]).max() < 3e-5 * S0[0], "singular values: %.2e" % (np.abs(Sk - S0[:k]).max() / S0[0])
            # (U is (n, ncomp) for 'lapack' and (ncomp, n) for every other mode: svd.py:597-606 -- ambiguous by shape when n == ncomp)
            Uk = U if mode == "lapack" else U.T
            assert Uk.shape[0] == n and Uk.shape[1] >= k, "U shape %s" % (U.shape,)
            rec = (Uk[:, :k] * Sk) @ V
            want = (U0[:, :k] * S0[:k]) @ V0[:k]
            if gap > 1e-3:
                assert np.abs(rec - want).max() < 2e-3 * max(1.0, np.abs(M).max() / 10) / min(1.0, gap * 50), "U S V: %.2e (gap %.1e; U %s)" % (np.abs(rec - want).max(), gap, U.shape)
        else:
            L = conv(out)
            assert L.shape == (n, k), "left vectors shape %s" % (L.shape,)
            assert np.abs(L.T @ L - np.eye(k)).max() < 2e-5, "left vectors not orthonormal"
            if gap > 1e-3:
                d = np.abs(L @ L.T - U0[:, :k] @ U0[:, :k].T).max()
                assert d < 5e-4 / min(1.0, gap * 50), "left subspace: %.2e (gap %.1e)" % (d, gap)
        print("ok   seed %d %s  (%.0f s so far)" % (seed, what, time.time() - t00), flush=True)
    except NotImplementedError as e:
        print("skip seed %d %s: %s" % (seed, what, str(e)[:100]), flush=True)
    except Exception as e:
        bad += 1
        print("FAIL seed %d %s: %s" % (seed, what, "".join(traceback.format_exception_only(type(e), e)).strip()[:400]), flush=True)
print("failures:", bad)
