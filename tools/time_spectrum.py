"""Whole spectrum + leading vectors (CEVR / svd_wrapper(full_output) of the eigen modes): vipmi_eigh_spectrum_f64 at n = 400."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi_device
ctx = B.get_context()
for kv in sys.argv[1:]:
    a, b = kv.split("="); ctx.set_option(a, int(b))
for n in (400, 200, 640, 1000):
    ct, _ = synth_adi_device(n, 128, seed=n); M = ct.reshape(n, -1); G = B.gram(M)
    w = np.linalg.eigvalsh(G.cpu().numpy())[::-1]
    for k, al in ((1, True), (20, True), (20, False)):
        B.eigh_topk(G.clone(), k, all_evals=al); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): ev, ec = B.eigh_topk(G.clone(), k, all_evals=al)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
        err = np.abs(ev.cpu().numpy()[:len(w) if al else k] - w[:len(w) if al else k]).max() / w[0]
        print("n=%d k=%d all_evals=%s: %.2f ms  (eigenvalue error %.1e)" % (n, k, al, dt, err), flush=True)
