import os, sys, cProfile, pstats, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi_device
ct, ang = synth_adi_device(400, 512, seed=0)
for _ in range(3):
    pca(ct, ang, ncomp=20, verbose=False, check_memory=False).cpu()
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); fr = pca(ct, ang, ncomp=20, verbose=False, check_memory=False); t1 = time.perf_counter()
    fr.cpu(); ts.append((t1 - t0) * 1e6)
print("pca() returns after (us):", sorted(ts)[:5], "median", sorted(ts)[10])
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    fr = pca(ct, ang, ncomp=20, verbose=False, check_memory=False); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
