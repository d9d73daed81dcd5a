"""Special matrices through every eigensolver entry: zero, identity, c * identity, rank one, a 2 x 2 block of equal eigenvalues,
1 x 1; sizes that select each solver family.  No hang, no error, eigenvalues right, unit vectors.   python tools/hunt_eigh_special.py"""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
bad = 0
rng = np.random.default_rng(3)
for n in (1, 2, 3, 17, 64, 130, 200, 257, 400, 520, 700, 1100):
    x = rng.standard_normal(n)
    mats = {"zero": np.zeros((n, n)), "identity": np.eye(n), "1e6 identity": 1e6 * np.eye(n), "rank one": np.outer(x, x),
            "1e-20 rank one": 1e-20 * np.outer(x, x), "two equal + noise": np.diag(np.concatenate([[7.0, 7.0][:min(2, n)], rng.uniform(0, 1, max(0, n - 2))]))}
    for name, G in mats.items():
        for k in sorted(set([1, min(n, 5), min(n, 64), n])):
            for batch in (1, 3):
                if batch > 1 and n > 300:
                    continue
                try:
                    Gt = torch.from_numpy(np.stack([G] * batch) if batch > 1 else G).cuda()
                    ev, ec = B.eigh_topk(Gt, k)
                    B.check_deferred()
                    ev = ev.cpu().numpy().reshape(batch, -1)[:, :k]; ec = ec.cpu().numpy().reshape(batch, -1, n)[:, :k]
                    w = np.linalg.eigvalsh(G)[::-1][:k]
                    scale = max(abs(w[0]), 1e-300)
                    assert np.isfinite(ev).all() and np.isfinite(ec).all(), "non-finite output"
                    assert np.abs(ev - w).max() <= 1e-11 * scale, "eigenvalues off by %.2e" % (np.abs(ev - w).max() / scale)
                    live = w > 1e-11 * scale
                    if live.any():
                        V = ec[0][live]
                        assert np.abs(V @ V.T - np.eye(int(live.sum()))).max() < 1e-9, "vectors not orthonormal"
                        assert np.abs(G @ V.T - V.T * ev[0][live]).max() < 1e-9 * scale, "residual"
                except Exception as e:
                    bad += 1
                    print("FAIL n %d %s k %d batch %d: %s" % (n, name, k, batch, "".join(traceback.format_exception_only(type(e), e)).strip()[:300]), flush=True)
                    try:
                        B.check_deferred()
                    except Exception:
                        pass
print("failures:", bad)
