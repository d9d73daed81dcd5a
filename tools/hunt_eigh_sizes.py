"""eigh_topk / eigh_topk(all_evals) / eigh at random sizes (n 2 .. 2300, batches 1 .. 40, k 1 .. n) against numpy: every solver family
and the boundaries between them (register-resident batched, LDS-streaming, cooperating multi-workgroup, wave-resident, large, Jacobi,
the verified fast path) on Gram matrices of noisy low-rank data, rank-deficient ones and clustered spectra.
   python tools/hunt_eigh_sizes.py [first [count]]"""
import sys, os, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bad = 0
t00 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(13000 + seed)
    kind = seed % 4
    if kind == 0:      # batches of small / medium problems
        n = int(rng.integers(2, 520)); batch = int(rng.integers(2, 40)) if n < 260 else int(rng.integers(1, 6))
    elif kind == 1:    # one problem, any size
        n = int(rng.integers(2, 2300)); batch = 1
    elif kind == 2:    # around the boundaries
        n = int(rng.choice([64, 96, 128, 129, 200, 201, 256, 257, 400, 512, 513, 640, 641, 1024, 2048, 2049])) + int(rng.integers(-1, 2)); batch = int(rng.choice([1, 1, 2, 8, 9]))
        n = max(n, 2)
        if n > 700:
            batch = 1
    else:
        n = int(rng.integers(30, 700)); batch = 1
    k = int(rng.integers(1, n + 1)) if rng.integers(3) == 0 else int(rng.integers(1, min(n, 64) + 1))
    shape = rng.integers(3)
    Gs = []
    for b in range(batch):
        if shape == 0:       # noisy low-rank data
            X = rng.standard_normal((n, 3 * n)); X[:, :min(5, 3 * n)] *= 10
        elif shape == 1:     # rank-deficient
            r = max(1, n // 3)
            X = rng.standard_normal((n, r))
        else:                # clustered spectrum
            Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
            lam = np.concatenate([np.full(n // 2, 5.0) + 1e-9 * rng.standard_normal(n // 2), rng.uniform(0.1, 1.0, n - n // 2)])
            X = Q * np.sqrt(lam)
        Gs.append(X @ X.T)
    G = np.stack(Gs)
    try:
        for mode in ("topk", "spectrum"):
            if mode == "spectrum" and (batch > 1 or n > 1200):
                continue
            Gt = torch.from_numpy(G if batch > 1 else G[0]).cuda()
            ev, ec = B.eigh_topk(Gt, k, all_evals=(mode == "spectrum"))
            B.check_deferred()
            ev = ev.cpu().numpy().reshape(batch, -1); ec = ec.cpu().numpy().reshape(batch, -1, n)
            for b in range(batch):
                w = np.linalg.eigvalsh(G[b])[::-1]
                scale = max(w[0], 1e-300)
                nev = n if mode == "spectrum" else k
                de = np.abs(ev[b, :nev] - w[:nev]).max() / scale
                assert de < 1e-11, "%s: eigenvalue error %.2e (problem %d)" % (mode, de, b)
                V = ec[b, :k]
                live = ev[b, :k] > 1e-11 * scale          # (rows of numerically null eigenvalues: unit vectors without meaning, include/vipmi.h)
                Vl = V[live]
                res = np.abs(G[b] @ Vl.T - Vl.T * ev[b, :k][live]).max() / scale if live.any() else 0.0
                orth = np.abs(Vl @ Vl.T - np.eye(int(live.sum()))).max() if live.any() else 0.0
                unit = np.abs((V * V).sum(1) - 1).max()
                assert res < 1e-9 and orth < 1e-9 and unit < 1e-9, "%s: residual %.2e orthogonality %.2e norms %.2e (problem %d)" % (mode, res, orth, unit, b)
        print("ok   seed %d n %d batch %d k %d spectrum shape %d  (%.0f s so far)" % (seed, n, batch, k, shape, time.time() - t00), flush=True)
    except Exception as e:
        bad += 1
        print("FAIL seed %d n %d batch %d k %d spectrum shape %d: %s" % (seed, n, batch, k, shape, "".join(traceback.format_exception_only(type(e), e)).strip()[:400]), flush=True)
print("failures:", bad)
