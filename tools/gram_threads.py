"""Concurrent B.gram() calls from several threads (own stream / context each): every result must equal the single-thread one."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi_device
OPTS = {o.split('=')[0]: int(o.split('=')[1]) for o in sys.argv[1:]}
ct, ang = synth_adi_device(400, 512, seed=0)
M = ct.reshape(400, -1)
def setopts():
    c = B.get_context()
    for a, b in OPTS.items(): c.set_option(a, b)
setopts()
ref = B.gram(M).clone()
errs = []
def work(k):
    with torch.cuda.stream(torch.cuda.Stream()):
        setopts()
        for i in range(40):
            G = B.gram(M)
            torch.cuda.current_stream().synchronize()
            if not torch.equal(G, ref): errs.append((k, i, float((G - ref).abs().max() / ref.abs().max())))
th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
[t.start() for t in th]; [t.join() for t in th]
print("opts", OPTS, "mismatches:", len(errs), errs[:4])
