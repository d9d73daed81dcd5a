"""Fast path of the leading-k eigensolver (csrc/eigh_chfsi.hip) against numpy on Gram matrices with known spectra: accuracy
(eigenvalues, residuals, principal angle of the k-dimensional subspace), the number of block products / Rayleigh-Ritz rounds,
the fall-back on hopeless spectra, and time against the exact tridiagonal path (option eigh_fast=0).

    python tools/eigh_fast_check.py [n k] ...
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vip_amd import backend as B


def gram_with_spectrum(n, lam, seed=0):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (Q * lam) @ Q.T


def baseline_like(n, k_sig=10, seed=0):
    """Spectrum of the BASELINE generator: a few geometric speckle modes above a narrow noise bulk (Marchenko-Pastur, n/P small)."""
    rng = np.random.default_rng(seed)
    bulk = 1.0 + 0.08 * np.sort(rng.uniform(-1, 1, n))[::-1] ** 3 + 0.06 * np.linspace(1, -1, n)
    lam = bulk * 0.024
    lam[:k_sig] += 1.0 * 2.0 ** (-np.arange(k_sig) * 1.2)
    return np.sort(lam)[::-1]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def check(name, G, k):
    n = G.shape[0]
    ctx = B.get_context()
    w, V = np.linalg.eigh(G)
    w, V = w[::-1], V[:, ::-1]
    Gt = torch.from_numpy(G).cuda()
    out = {}
    for fast in (1, 0):
        ctx.set_option("eigh_fast", fast)
        ev, ec = B.eigh_topk(Gt.clone(), k)
        torch.cuda.synchronize()
        ev, ec = ev.cpu().numpy(), ec.cpu().numpy()
        res = np.linalg.norm(G @ ec.T - ec.T * ev, axis=0).max() / w[0]
        sin = np.linalg.svd(V[:, k:].T @ ec.T, compute_uv=False).max()
        orth = np.abs(ec @ ec.T - np.eye(k)).max()
        info = [ctx.get_option("eigh_fast_last_" + s) for s in ("products", "rounds", "locked", "reason")] if fast else None
        ms = timed(lambda: B.eigh_topk(Gt.clone(), k))
        out[fast] = (np.abs(ev - w[:k]).max() / w[0], res, sin, orth, ms, info)
    ctx.set_option("eigh_fast", 1)
    f, e = out[1], out[0]
    print("%-26s n=%4d k=%2d | fast: dlam %.1e res %.1e sin %.1e orth %.1e %7.3f ms info(products, rounds, locked, reason)=%s | "
          "exact: dlam %.1e res %.1e sin %.1e %7.3f ms" % (name, n, k, f[0], f[1], f[2], f[3], f[4], f[5], e[0], e[1], e[2], e[4]))
    return out


if __name__ == "__main__":
    torch.cuda.set_device(0)
    if len(sys.argv) > 1 and sys.argv[1] == "prof":            # two cases, fast path only (for rocprofv3 --kernel-trace --stats)
        ctx = B.get_context()
        cases = ((400, 20, 3), (2000, 50, 3)) if len(sys.argv) < 3 else ((int(sys.argv[2]), int(sys.argv[3]), 0),)
        for n, k, sd in cases:
            Gt = torch.from_numpy(gram_with_spectrum(n, baseline_like(n, seed=sd))).cuda()
            for _ in range(6):
                B.eigh_topk(Gt.clone(), k)
            torch.cuda.synchronize()
            print(n, k, [ctx.get_option("eigh_fast_last_" + s) for s in ("products", "rounds", "locked", "reason")])
        sys.exit(0)
    if os.path.exists("/tmp/proto/G_c2.npy"):
        check("C2 generator (file)", np.load("/tmp/proto/G_c2.npy"), 20)
    from vip_amd.synth import synth_adi
    cube, _ = synth_adi(400, 128, seed=0)
    M = cube.reshape(400, -1).astype(np.float64)
    check("generator 400x128^2", M @ M.T, 20)
    check("baseline-like 400", gram_with_spectrum(400, baseline_like(400)), 20)
    check("baseline-like 400 k=10", gram_with_spectrum(400, baseline_like(400)), 10)
    check("geometric 0.9^i", gram_with_spectrum(400, 0.9 ** np.arange(400)), 20)
    check("power law i^-1.5", gram_with_spectrum(512, (1.0 + np.arange(512)) ** -1.5), 30)
    check("flat bulk (hopeless)", gram_with_spectrum(400, 1.0 + 1e-4 * np.linspace(1, 0, 400)), 20)
    check("repeated top pairs", gram_with_spectrum(400, np.r_[np.repeat([5.0, 3.0, 2.0], 4), 0.5 * 0.97 ** np.arange(388)]), 16)
    check("rank 30 (zeros below)", gram_with_spectrum(400, np.r_[2.0 ** -np.arange(30.0), np.zeros(370)]), 20)
    check("n=333 odd", gram_with_spectrum(333, baseline_like(333)), 17)
    check("baseline-like 1000", gram_with_spectrum(1000, baseline_like(1000)), 30)
    check("baseline-like 2000 k=50", gram_with_spectrum(2000, baseline_like(2000, seed=3)), 50)
