"""The building-block kernels at random sizes against numpy (no oracle needed: the definitions are one line each): Gram matrices
(int8 digit planes / float64 MFMA / batched, odd row lengths and leading dimensions), the truncated projection (residuals of M with
respect to the top-k PCs of M or of a reference), matrix scalings, masks, every collapse with NaNs (median bit-exact).
   python tools/hunt_kernels.py [first [count]]"""
import sys, os, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bad = 0
t00 = time.time()


def up(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


for seed in range(first, first + count):
    rng = np.random.default_rng(17000 + seed)
    kind = seed % 5
    what = ""
    try:
        if kind == 0:                                           # Gram
            n = int(rng.integers(1, 700)); P = int(rng.integers(1, 300000)) if rng.integers(3) else int(rng.integers(1, 3000))
            if n * P > 6e7:
                P = int(6e7 // n)
            ld = P + int(rng.integers(0, 9)) * int(rng.integers(2))
            M = (rng.standard_normal((n, ld)) * rng.uniform(0.01, 100)).astype(np.float32)
            if rng.integers(4) == 0:
                M[:, : max(1, P // 7)] += 1000.0                  # an offset: many digits at once
            what = "gram n %d P %d ld %d" % (n, P, ld)
            Mt = up(M)
            G = B.gram(Mt[:, :P]).cpu().numpy()
            M64 = M[:, :P].astype(np.float64)
            Gw = M64 @ M64.T
            err = np.abs(G - Gw).max() / max(np.abs(Gw).max(), 1e-300)
            assert err < 2e-11, "relative error %.2e" % err
        elif kind == 1:                                         # batched Gram
            b = int(rng.integers(1, 45)); n = int(rng.integers(1, 260)); P = int(rng.integers(1, 70000))
            if b * n * P > 5e7:
                P = max(1, int(5e7 // (b * n)))
            M = rng.standard_normal((b, n, P)).astype(np.float32)
            what = "gram_batched b %d n %d P %d" % (b, n, P)
            G = B.gram_batched(up(M)).cpu().numpy()
            M64 = M.astype(np.float64)
            Gw = M64 @ M64.transpose(0, 2, 1)
            err = np.abs(G - Gw).max() / max(np.abs(Gw).max(), 1e-300)
            assert err < 2e-11, "relative error %.2e" % err
        elif kind == 2:                                         # projection
            n = int(rng.integers(2, 300)); P = int(rng.integers(n + 1, 60000)); k = int(rng.integers(1, min(n, 64) + 1))
            X = rng.standard_normal((n, P)).astype(np.float32)
            X[:, : P // 3] += (rng.standard_normal((n, 1)) * rng.standard_normal((1, P // 3)) * 5).astype(np.float32)
            ref = None
            if rng.integers(2):
                nr = int(rng.integers(k, 200))
                ref = rng.standard_normal((nr, P)).astype(np.float32)
            what = "pca_project n %d P %d k %d ref %s" % (n, P, k, None if ref is None else ref.shape[0])
            res = B.pca_project(up(X), k, None if ref is None else up(ref))[0].cpu().numpy()
            R = (X if ref is None else ref).astype(np.float64)
            w, v = np.linalg.eigh(R @ R.T)
            E = v[:, ::-1][:, :k]
            gap = (w[::-1][k - 1] - (w[::-1][k] if k < R.shape[0] else 0.0)) / w[-1]
            V = E.T @ R
            V /= np.linalg.norm(V, axis=1, keepdims=True)
            want = X.astype(np.float64) - (X.astype(np.float64) @ V.T) @ V
            err = np.abs(res - want).max()
            tol = 5e-5 * max(1.0, np.abs(X).max() / 10) / min(1.0, max(gap * 1e4, 1e-3))
            assert err < tol, "max|d| %.2e (relative gap at k %.1e)" % (err, gap)
        elif kind == 3:                                         # scalings and masks
            n = int(rng.integers(1, 400)); P = int(rng.integers(1, 90000))
            M = (rng.standard_normal((n, P)) * rng.uniform(0.1, 30) + rng.uniform(-50, 50)).astype(np.float32)
            if rng.integers(3) == 0 and P > 3:
                M[:, 1] = 7.0                                     # a constant column: sd -> 1
            mode = ("temp-mean", "spat-mean", "temp-standard", "spat-standard")[rng.integers(4)]
            what = "scale %s n %d P %d" % (mode, n, P)
            out = B.scale(up(M), mode).cpu().numpy()
            ax = 0 if mode.startswith("temp") else 1
            M64 = M.astype(np.float64)
            want = M64 - M64.mean(ax, keepdims=True)
            if mode.endswith("standard"):
                sd = M64.std(ax, keepdims=True)
                sd[sd < 10 * np.finfo(np.float32).eps] = 1.0
                want = want / sd
            err = np.abs(out - want).max()
            assert err < 3e-5 * max(1.0, np.abs(want).max()), "max|d| %.2e" % err
        else:                                                   # collapses
            n = int(rng.integers(1, 900)); P = int(rng.integers(1, 40000))
            if n * P > 4e7:
                P = int(4e7 // n)
            C = rng.standard_normal((n, P)).astype(np.float32)
            if rng.integers(2):
                C[rng.random((n, P)) < rng.uniform(0, 0.3)] = np.nan
                C[:, :: max(2, P // 5)] = np.nan                  # whole-NaN pixels
            if rng.integers(3) == 0:
                C = np.round(C * 2) / 2                           # many ties
            what = "collapse n %d P %d" % (n, P)
            Ct = up(C).reshape(n, 1, P)
            with np.errstate(all="ignore"):
                import warnings
                warnings.simplefilter("ignore")
                for mode, fn in (("median", np.nanmedian), ("mean", np.nanmean), ("sum", np.nansum), ("max", np.nanmax)):
                    got = B.collapse(Ct, mode).cpu().numpy().reshape(P)
                    want = fn(C.astype(np.float64) if mode != "median" else C, axis=0)
                    if mode == "sum":
                        pass
                    ok = np.isfinite(want)
                    if mode == "median":
                        assert np.array_equal(np.isnan(got), np.isnan(want)), "median NaN pattern"
                        assert np.array_equal(got[ok], want[ok].astype(np.float32)), "median not bit-exact (max|d| %.2e)" % np.abs(got[ok] - want[ok]).max()
                    else:
                        if mode != "sum":
                            assert np.array_equal(np.isnan(got), np.isnan(want)), mode + " NaN pattern"
                        d = np.abs(got[ok] - want[ok]).max() if ok.any() else 0.0
                        assert d < 1e-5 * max(1, n) ** 0.5 * 4, "%s max|d| %.2e" % (mode, d)
        print("ok   seed %d %s  (%.0f s so far)" % (seed, what, time.time() - t00), flush=True)
    except Exception as e:
        bad += 1
        print("FAIL seed %d %s: %s" % (seed, what, "".join(traceback.format_exception_only(type(e), e)).strip()[:400]), flush=True)
print("failures:", bad)
