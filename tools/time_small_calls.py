"""Per-call time of vipmi_pca_fullframe_f32 through raw ctypes on small cubes (the NEGFC / contrast-curve regime):
wall time per call of a back-to-back loop on one stream against the sum of the kernel times (stage timers)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi
for n, N, k in ((50, 128, 5), (30, 64, 3), (100, 101, 10)):
    cube_np, ang = synth_adi(n, N, 0)
    ctx = B.get_context()
    cube = torch.from_numpy(cube_np).cuda(); frame = torch.empty((N, N), device="cuda")
    angles = np.ascontiguousarray(ang, dtype=np.float64)
    call = lambda: ctx.call("vipmi_pca_fullframe_f32", B.ptr(cube), angles.ctypes.data, n, N, k, 0, None, 0, B.ptr(frame), None, None, None, None)
    for _ in range(5): call()
    torch.cuda.synchronize()
    reps = 200
    t = time.perf_counter()
    for _ in range(reps): call()
    t_issue = (time.perf_counter() - t) / reps
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t) / reps
    ctx.set_option("timing", 1); ctx.reset_timers()
    for _ in range(20): call()
    torch.cuda.synchronize()
    st = {s: round(ctx.stage_ms(s) / 20 * 1e3, 1) for s in ("scale", "gram", "eigh", "project", "derotate", "collapse")}
    ctx.set_option("timing", 3); ctx.reset_timers()
    for _ in range(20): call()
    torch.cuda.synchronize()
    hs = {s: round(ctx.stage_ms(s) / 20 * 1e3, 1) for s in ("scale", "gram", "eigh", "project", "derotate", "collapse")}
    ctx.set_option("timing", 0)
    print("   host time inside each stage (us; includes waiting for a free staging slot when the host runs ahead of the GPU):", hs, "sum %.1f" % sum(v for v in hs.values() if v > 0))
    print("%dx%dx%d k=%d: %.1f us per call back to back (host issue %.1f us); stage kernels (us): %s sum %.1f" % (
        n, N, N, k, t_all * 1e6, t_issue * 1e6, st, sum(v for v in st.values() if v > 0)))
