"""Gram accumulation variants at C2 / C5 scale: time of the Gram stage and the change of the PCA results against the
float64-MFMA default (max |d frame|, max |d residuals|, ||sin Theta||_2 of the PC subspaces).  Run on the GPU box;
variants through VIPMI_OPTS (gram_f32=1[,gram_slices=N])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi, synth_adi_device

variant = os.environ.get("VIPMI_OPTS", "")


def run(cube_t, ang, k):
    ctx = B.get_context()
    ctx.set_option("timing", 1)
    out = pca(cube_t, ang, ncomp=k, full_output=True, verbose=False, check_memory=False)
    ctx.reset_timers()
    for _ in range(3):
        pca(cube_t, ang, ncomp=k, verbose=False, check_memory=False)
    torch.cuda.synchronize()
    return out, ctx.stage_ms("gram") / 3


def subspace_sin(Va, Vb):
    Va, Vb = Va.double(), Vb.double()
    D = Va - (Va @ Vb.T) @ Vb
    return float(torch.linalg.eigvalsh(D @ D.T)[-1].clamp(min=0).sqrt())


for tag, gen, k in (("C2 400x512x512 k=20", lambda: (torch.from_numpy(synth_adi(400, 512, 0)[0]).cuda(), np.linspace(0, 90, 400)), 20),
                    ("C5 2000x1024x1024 k=50", lambda: synth_adi_device(2000, 1024, 0), 50)):
    cube_t, ang = gen()
    (fr, pcs, rec, res, der), t_var = run(cube_t, ang, k)
    fr, pcs, res = fr.clone(), pcs.reshape(k, -1).clone(), res.clone()
    del rec, der
    os.environ["VIPMI_OPTS"] = ""
    B.release_workspaces()
    ctx = B.get_context()
    ctx.set_option("gram_f32", 0)
    (fr0, pcs0, rec0, res0, der0), t_ref = run(cube_t, ang, k)
    print("%s  variant[%s]: gram %.3f ms (f64 default %.3f ms)  max|d frame| %.2e  max|d residuals| %.2e  sin(theta) %.2e"
          % (tag, variant, t_var, t_ref, float((fr - fr0).abs().max()), float((res - res0).abs().max()),
             subspace_sin(pcs, pcs0.reshape(k, -1))))
    del cube_t, fr0, pcs0, rec0, res0, der0, fr, pcs, res
    os.environ["VIPMI_OPTS"] = variant
    B.release_workspaces()
    torch.cuda.empty_cache()
