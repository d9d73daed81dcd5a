"""Soak of the verified fast eigensolver path: many back-to-back calls on one 2000 x 2000 Gram matrix (and on a 700 x 700
one) must return bit-identical pairs every time, interleaved with calls on other sizes (workspace reuse).
    python tools/soak_fast_eigh.py [calls]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vip_amd import backend as B
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from eigh_fast_check import gram_with_spectrum, baseline_like

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = B.get_context()
mats = {(n, k): torch.from_numpy(gram_with_spectrum(n, baseline_like(n, seed=n))).cuda() for n, k in ((2000, 50), (700, 20), (1000, 30))}
first = {}
bad = 0
for i in range(calls):
    for (n, k), G in mats.items():
        ev, ec = B.eigh_topk(G.clone(), k)
        assert ctx.get_option("eigh_fast_last_reason") == 0
        if (n, k) not in first:
            first[(n, k)] = (ev.clone(), ec.clone())
        elif not (torch.equal(ev, first[(n, k)][0]) and torch.equal(ec, first[(n, k)][1])):
            bad += 1
torch.cuda.synchronize()
print("soak: %d calls per matrix, %d results differing from the first" % (calls, bad))
assert bad == 0
