"""Small driver used under rocprofv3: runs one stage of the pipeline a few times on synthetic data.
usage: python tools/prof_stage.py {derotate|eigh|gram|pca} [n] [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi

what = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
N = int(sys.argv[3]) if len(sys.argv) > 3 else 512
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
cube, ang = synth_adi(n, N, seed=0, planet=False)
ct = torch.from_numpy(cube).cuda()
ctx = B.get_context()
ctx.set_option("timing", 1)
for kv in os.environ.get("VIPMI_OPTS", "").split(","):
    if "=" in kv:
        k_, v_ = kv.split("="); ctx.set_option(k_, int(v_))
for r in range(reps):
    ctx.reset_timers()
    if what == "derotate":
        B.derotate(ct, ang)
    elif what == "gram":
        B.gram(ct.reshape(n, -1))
    elif what == "eigh":
        G = B.gram(ct.reshape(n, -1))
        B.eigh(G)
    elif what == "pca":
        from vip_amd.psfsub import pca
        pca(ct, ang, ncomp=20, verbose=False, check_memory=False)
    torch.cuda.synchronize()
    print(what, "rep", r, {s: round(ctx.stage_ms(s), 3) for s in ("gram", "eigh", "project", "derotate", "collapse", "k_rot_s1", "k_rot_s2", "k_rot_s3") if ctx.stage_count(s)},
          "sweeps", ctx.get_option("eigh_last_sweeps"))
