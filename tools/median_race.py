import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi_device
OPTS = {o.split('=')[0]: int(o.split('=')[1]) for o in sys.argv[1:]}
LOPTS = {k[1:]: v for k, v in OPTS.items() if k.startswith('L')}        # L<option>=v: applied to the loader threads' contexts only
OPTS = {k: v for k, v in OPTS.items() if not k.startswith('L')}
ct, ang = synth_adi_device(400, 512, seed=0)
M = ct.reshape(400, -1)
c0 = B.get_context()
for a, b in OPTS.items(): c0.set_option(a, b)
ref = B.collapse(ct, "median").clone(); torch.cuda.synchronize()
refm = torch.median(ct, dim=0).values if ct.shape[0] % 2 else None
stop = [False]
def loader():
    with torch.cuda.stream(torch.cuda.Stream()):
        for a_, b_ in LOPTS.items(): B.get_context().set_option(a_, b_)
        while not stop[0]:
            B.gram(M); torch.cuda.current_stream().synchronize()
tl = [threading.Thread(target=loader) for _ in range(2)]; [t.start() for t in tl]
with torch.cuda.stream(torch.cuda.Stream()):
    c = B.get_context()
    for a, b in OPTS.items(): c.set_option(a, b)
    for i in range(6):
        o = B.collapse(ct, "median"); torch.cuda.current_stream().synchronize()
        d = (o - ref).abs()
        bad = (d > 0).nonzero()
        print("call %d: %d pixels differ, max %.3g, first rows/cols %s" % (i, bad.shape[0], float(d.max()), bad[:6].tolist()))
stop[0] = True; [t.join() for t in tl]
