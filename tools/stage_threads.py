"""Which pair of stages interferes across threads?  One thread loops stage X, another stage Y (own stream / context each); the
checked stage must reproduce its single-thread result.   python tools/stage_threads.py"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi_device
ct, ang = synth_adi_device(400, 512, seed=0)
M = ct.reshape(400, -1)
G0 = B.gram(M).clone()
def st_gram(): return B.gram(M)
def st_eigh():
    ev, ec = B.eigh_topk(G0.clone(), 20); return torch.cat([ev.flatten(), ec.flatten()])
def st_proj(): return B.pca_project(M, 20)[0]
def st_rot(): return B.derotate(ct, ang)
def st_med(): return B.collapse(ct, "median")
def st_pca(): return pca(ct, ang, ncomp=20, verbose=False, check_memory=False)
S = {"gram": st_gram, "eigh": st_eigh, "project": st_proj, "derotate": st_rot, "median": st_med, "pca": st_pca}
OPTS = {o.split("=")[0]: int(o.split("=")[1]) for o in sys.argv[1:] if "=" in o}
sys.argv = [a for a in sys.argv if "=" not in a]
pairs = [tuple(a.split("+")) for a in sys.argv[1:]] or [("eigh", "gram"), ("project", "gram"), ("derotate", "gram"), ("median", "gram"), ("pca", "gram"), ("pca", "pca")]
for chk, load in pairs:
    ref = S[chk]().clone(); torch.cuda.synchronize()
    stop, errs = [False], []
    def loader():
        with torch.cuda.stream(torch.cuda.Stream()):
            for a_, b_ in OPTS.items(): B.get_context().set_option(a_, b_)
            while not stop[0]:
                S[load](); torch.cuda.current_stream().synchronize()
    def checker():
        with torch.cuda.stream(torch.cuda.Stream()):
            for i in range(40):
                o = S[chk](); torch.cuda.current_stream().synchronize()
                same = torch.equal(torch.nan_to_num(o, nan=1234.5), torch.nan_to_num(ref, nan=1234.5))
                if not same: errs.append(i)
    tl = [threading.Thread(target=loader) for _ in range(2)]; tc = threading.Thread(target=checker)
    [t.start() for t in tl]; tc.start(); tc.join(); stop[0] = True; [t.join() for t in tl]
    print("checked %-9s under 2 threads of %-9s: %d of 40 calls differ" % (chk, load, len(errs)))
