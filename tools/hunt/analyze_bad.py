"""What do the wrong medians look like?  (unmodified library; loaders = int8 Gram on two other streams)"""
import os, sys, threading, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi_device
n, N = 400, 512
ct, ang = synth_adi_device(n, N, seed=0)
M = ct.reshape(n, -1)
srt = torch.sort(M, dim=0).values
ref = ((srt[199] + srt[200]) * 0.5)
torch.cuda.synchronize()
stop = [False]
def loader():
    with torch.cuda.stream(torch.cuda.Stream()):
        while not stop[0]:
            B.gram(M); torch.cuda.current_stream().synchronize()
tl = [threading.Thread(target=loader) for _ in range(2)]; [t.start() for t in tl]
import time; time.sleep(0.3)
outs = []
with torch.cuda.stream(torch.cuda.Stream()):
    for i in range(8):
        o = B.collapse(ct, "median"); torch.cuda.current_stream().synchronize()
        outs.append(o.flatten().clone())
stop[0] = True; [t.join() for t in tl]
S = srt.cpu().numpy()
R = ref.cpu().numpy()
for ci, o in enumerate(outs):
    o = o.cpu().numpy()
    bad = np.nonzero(~((o == R) | (np.isnan(o) & np.isnan(R))))[0]
    if not bad.size:
        print("call", ci, "clean"); continue
    j = bad % 16
    tiles = bad // 16
    tc = collections.Counter(tiles.tolist())
    print("call %d: %d bad px in %d tiles; px per bad tile hist %s; j hist %s" % (
        ci, bad.size, len(tc), dict(collections.Counter(tc.values())), dict(sorted(collections.Counter(j.tolist()).items()))))
    # tile id -> blockIdx.x (xcd_ranges mapping: tile = (bid & 7) * per_xcd + (bid >> 3)); per_xcd = 2048
    per = 16384 // 8
    bids = [(t % per) * 8 + (t // per) for t in tc]
    print("   tile ids (first 20)", sorted(tc)[:20])
    print("   xcd of bad tiles", dict(sorted(collections.Counter([t // per for t in tc]).items())))
    print("   position inside the XCD range (//128)", dict(sorted(collections.Counter([(t % per) // 128 for t in tc]).items())))
    kinds = collections.Counter()
    detail = []
    for p in bad[:400]:
        v = o[p]; s = S[:, p]
        if np.isnan(v): kinds["nan"] += 1; continue
        # v == (s[a] + s[b]) / 2 for some a <= b ?
        hit = None
        two = np.float32(2) * v
        for a in range(n):
            bval = np.float32(two - s[a])
            idx = np.searchsorted(s, bval)
            for b in (idx - 1, idx, idx + 1):
                if 0 <= b < n and np.float32((s[a] + s[b]) * np.float32(0.5)) == v:
                    hit = (a, b); break
            if hit: break
        if hit is None: kinds["not a pair mean"] += 1; detail.append((int(p), float(v), float(R[p])))
        else:
            kinds["pair(%+d,%+d)" % (hit[0] - 199, hit[1] - 200)] += 1
    print("   kinds:", dict(kinds.most_common(12)))
    print("   not-pair samples:", detail[:5])
    # neighbours: does the wrong value equal another pixel's correct median?
    eq_other = 0
    rs = set(R.tolist())
    for p in bad[:400]:
        if float(o[p]) in rs: eq_other += 1
    print("   of first 400 bad: equal to SOME pixel's correct median:", eq_other)
